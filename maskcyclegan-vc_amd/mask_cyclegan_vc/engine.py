"""Hand-scheduled MaskCycleGAN-VC training step on the HIP library (the product's fast path).

Semantics = reference ``MaskCycleGANVCTraining.train()`` inner loop (train.py:195-299): 6 generator
and 4 discriminator forwards + backward + Adam(G), then 4+8 discriminator forwards on real /
generated data (generators run with their *updated* weights) + backward + Adam(D).

What differs from running the reference loop on autograd -- all mathematically invisible:
  * no autograd graph: the step's dataflow is fixed, so forward/backward library calls are issued
    directly in dependency order with pre-allocated buffers (graph-capturable, no allocation);
  * work whose result the reference throws away is not computed: discriminator weight gradients in
    the generator phase (zeroed by ``reset_grad`` at train.py:297 before use) and the backward
    through the generators in the discriminator phase (zeroed at :240 of the next iteration);
  * parameters, gradients and Adam moments live in flat buffers (generator: both nets; discriminator:
    the live tensors of all four nets -- ``downSample4`` never receives a gradient, so like
    torch.optim.Adam we never touch it), giving one Adam launch and one all-reduce per phase.
The modules' ``nn.Parameter``s are re-pointed at views of the flat buffers, so ``state_dict()``,
checkpoints and the autograd API keep working on the same storage.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _hip
from ._hip import check, lib, ptr, ptr_table, stream
from .lanes import get_lanes
from .optim_state import load_optimizer_state_dict, optimizer_state_dict
from .parallel import FlatGradReducer
from .schedule import StepSchedule

G_NAMES = ("generator_A2B", "generator_B2A")
D_NAMES = ("discriminator_A", "discriminator_B", "discriminator_A2", "discriminator_B2")
_DEAD = range(14, 18)          # discriminator downSample4.* slots in named_parameters() order

# loss slots (device float32[16])
# loss slots: one block per phase (the discriminator phase's tail may still be writing its block while the next iteration's generator
# phase zeroes its own): [public 4 | 8 private (weighted value, mean) pairs]
SLOT_G, SLOT_CYCLE, SLOT_IDENT, SLOT_ADV_G = range(4)
SLOT_D, SLOT_D_REAL, SLOT_D_FAKE = 20, 21, 22
_BLOCK = 20


def _pair(k):
    """Offset of the private pair of loss call k (0..7 generator phase, 8..15 discriminator phase)."""
    return 4 + 2 * k if k < 8 else _BLOCK + 4 + 2 * (k - 8)


def _align4(n):
    return (n + 3) & ~3


class _FlatGroup:
    """Parameters of several modules re-homed into one flat buffer (+ gradient and Adam buffers)."""

    def __init__(self, param_lists, device):
        sizes = [[p.numel() for p in ps] for ps in param_lists]
        total = sum(_align4(n) for ss in sizes for n in ss)
        self.flat = torch.zeros(total, device=device)
        self.grad = torch.zeros(total, device=device)
        self.exp_avg = torch.zeros(total, device=device)
        self.exp_avg_sq = torch.zeros(total, device=device)
        self.step = 0
        self.views, self.grad_views = [], []
        off = 0
        for ps in param_lists:
            vs, gs = [], []
            for p in ps:
                n = p.numel()
                v = self.flat[off:off + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = self.grad[off:off + n].view(p.shape)
                p.grad = g
                vs.append(v); gs.append(g)
                off += _align4(n)
            self.views.append(vs); self.grad_views.append(gs)
        self.numel = total
        self.offsets = []                 # flat-buffer offset of every tensor, per module
        off = 0
        for ps in param_lists:
            o = []
            for p in ps:
                o.append(off)
                off += _align4(p.numel())
            self.offsets.append(o)


class TrainEngine:
    def __init__(self, nets, batch_size, n_frames=64, schedule: StepSchedule | None = None, reducer: FlatGradReducer | None = None,
                 betas=(0.5, 0.999), eps=1e-8):
        self.nets = nets
        dev = next(nets[G_NAMES[0]].parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("TrainEngine needs the networks on a HIP device (no CPU path)")
        self.device = dev
        self.B, self.T = batch_size, n_frames
        self.sched = schedule or StepSchedule(batch_size=batch_size, n_samples=batch_size)   # (the reference divides by n_samples // batch_size)
        self.reducer = reducer or FlatGradReducer()
        self.betas, self.eps = betas, eps
        L = self.L = lib()
        B, T = self.B, self.T
        # ---- flat parameter groups
        g_lists = [list(nets[n].parameters()) for n in G_NAMES]
        d_all = [list(nets[n].parameters()) for n in D_NAMES]
        d_live = [[p for i, p in enumerate(ps) if i not in _DEAD] for ps in d_all]
        self.g_group = _FlatGroup(g_lists, dev)
        self.d_group = _FlatGroup(d_live, dev)
        self._p_tab, self._g_tab = {}, {}
        for n, ps, gv in zip(G_NAMES, g_lists, self.g_group.grad_views):
            self._p_tab[n] = ptr_table(ps)
            self._g_tab[n] = ptr_table(gv)
        for n, ps, gv in zip(D_NAMES, d_all, self.d_group.grad_views):
            self._p_tab[n] = ptr_table(ps)
            it = iter(gv)
            self._g_tab[n] = ptr_table([None if i in _DEAD else next(it) for i in range(len(ps))])
        # optimizer.step() is fused with the weight re-pack (mcvc_gen_update_ranges / mcvc_disc_update_batch, r4): ONE launch per network (pair)
        # updates a tile of filters, keeps it in LDS and writes every packed copy from there -- no per-step `pack` launches, no second read
        # of the OIHW tensors, forward AND backward copies fresh when the step returns.  Bit-identical to Adam + re-pack (tests/test_hip_update.py).
        ll = lambda v: (ctypes.c_longlong * len(v))(*v)     # noqa: E731
        self._numel = {n: ll([p.numel() for p in ps]) for n, ps in zip(G_NAMES, g_lists)}
        self._numel.update({n: ll([0 if i in _DEAD else p.numel() for i, p in enumerate(ps)]) for n, ps in zip(D_NAMES, d_all)})
        # flat-buffer range of each discriminator's live parameters (the discriminator pairs are updated separately when pipelined)
        self._d_ranges = {}
        d_off = [o[0] for o in self.d_group.offsets] + [self.d_group.numel]
        for i, n in enumerate(D_NAMES):
            self._d_ranges[n] = (d_off[i], d_off[i + 1])
        # ---- packed weights
        self.packed = {n: torch.zeros(L.mcvc_gen_packed_floats(), device=dev) for n in G_NAMES}
        self.packed.update({n: torch.zeros(L.mcvc_disc_packed_floats(), device=dev) for n in D_NAMES})
        # the loss calls of a phase run on different lanes, each into its private pair; mcvc_loss_combine adds them to the public slots
        # in the reference's order
        self.slots = torch.zeros(2 * _BLOCK, device=dev)
        ia = lambda v: (ctypes.c_int * len(v))(*v)          # noqa: E731
        self._comb_g = (8, ia([SLOT_G] * 8), ia([SLOT_CYCLE, SLOT_CYCLE, SLOT_IDENT, SLOT_IDENT] + [SLOT_ADV_G] * 4))
        self._comb_d = (8, ia([SLOT_D] * 8), ia([SLOT_D_REAL, SLOT_D_FAKE] * 4))
        # The A->B and B->A halves of the step are independent chains over different networks.  At small batch their
        # kernels are latency-bound and far from filling 256 CUs, so the two chains run as two "lanes" on two HIP
        # streams (lane 0 = the caller's stream) and overlap on the chip; join points are stream-event waits.
        self.concurrent = True
        self._serial = False                   # bench / tools: submit the task graphs of the CURRENT schedule in order on one stream (per-kernel tracing)
        # lanes: the caller's stream + side streams on distinct hardware queues (lanes.py)
        self._sides_legacy, self._aux, self._sides_probed, self.queue_probe = get_lanes(dev)
        self.aux_wgrad = True                  # generators' weight gradients on auxiliary streams ...
        self.aux_wgrad_d = False               # ... not the discriminators': their four lanes occupy the four hardware queues (1.2 % slower)
        # (HIP-graph replay of a phase / of a network pass: built in rounds 1-2, measured slower or equal on ROCm 7.2, removed in r5 -- DESIGN 5.)
        self._task_events = {}
        self._g_fwd_packed = False
        self._g_grad_clean = self._d_grad_clean = False
        # Grouped launches (csrc/twin.h): G_A2B / G_B2A -- and the discriminator pairs -- run the same layer schedule on different weights,
        # so every kernel of a pair of passes goes out ONCE with gridDim.z = 2 instead of twice on two lanes: half the launches, twice the
        # workgroups per launch, the two chains in lock-step.  ``grouped_max_b``: largest per-pass batch for which the grouped schedule is
        # used (from 8 samples per pass the kernels fill the chip per network and the four-lane schedule is as fast: DESIGN section 5).
        self.grouped = True
        self.grouped_max_b = 4
        # Pipelined step (needs the grouped schedule): iteration t's discriminator phase is issued together with iteration t+1's
        # generator phase, as one task graph (_pipelined_step); False keeps the two phases of an iteration back to back (tests).
        self.pipelined = True
        # Merged forwards (r4): the discriminator phase's generator forwards of iteration t (train.py:259-273) read the same generator
        # weights as the generator phase's forwards of iteration t+1 (:203-210) and need no gradient: they ride in those passes as one
        # more sample (_merged_step).  Slower than the separate passes on one GPU (6.20-6.26 vs 6.00-6.16 ms at bs=1: DESIGN section 5);
        # default on data-parallel ranks only -- there one grouped persistent trunk pass in flight instead of two leaves half the compute
        # units to RCCL's kernels (csrc/trunk.h residency rule).
        self.merged = self.reducer.active
        self.trunk_fallback = False            # a persistent trunk launch faulted in this process: per-layer launches from then on (check_faults)
        self._pending_D = None                  # (input set, discriminator lr) of the iteration whose discriminator phase is still to run
        self.slots_done = torch.zeros(2 * _BLOCK, device=dev)          # loss slots of the last COMPLETE iteration
        self._done_ring = [(torch.zeros(2 * _BLOCK).pin_memory(), torch.cuda.Event()) for _ in range(4)]    # published loss slots (host) + events
        self._done_count = 0                    # iterations published so far ...
        self._done_base = 0                     # ... of which before the last synchronous read (losses() without a lag)
        self.split_d_min_batch = 8                  # (r5 same-box A/B: bs=4 12.94 -> 12.85 ms unsplit, bs=8 21.42 split vs 21.52)
        self._timeline = None
        # data parallel (plain schedules): start the discriminator gradient all-reduce at the end of an iteration and finish the update
        # where the discriminators are next used, i.e. after the next generator forwards -- the exchange hides behind them.  On one GPU
        # the same deferral measured slower (6.91 -> 7.26 ms at bs=1: the update's 0.7 GB of traffic beside the generator forwards).
        self.defer_d_update = self.reducer.active
        self._pending_d_lr = None
        # ... and the generator gradient all-reduce starts per parameter range while the last backward passes are still
        # running: the library records an event when a range's gradients are complete (mcvc_gen_backward_overlap)
        self.overlap_g_reduce = self.reducer.active and dev.type == "cuda"
        self._ms = {}
        for n in G_NAMES:
            evs = [torch.cuda.Event() for _ in range(4)]        # ([2], [3]: the head's finer milestones, MCVC_BWD_FINE_MILESTONES)
            for e in evs:
                e.record()                                  # forces creation of the underlying hipEvent_t
            self._ms[n] = (evs, (ctypes.c_void_p * 4)(*[e.cuda_event for e in evs]))
        # flat-buffer ranges [lo, hi) of parameters [100,110), [24,100), [0,24) of each generator
        self._g_ranges = {}
        base = self.g_group.grad.data_ptr()
        for n, gv in zip(G_NAMES, self.g_group.grad_views):
            off = [(g.data_ptr() - base) // 4 for g in gv] + [(gv[-1].data_ptr() - base) // 4 + _align4(gv[-1].numel())]
            self._g_ranges[n] = [(off[100], off[110]), (off[24], off[100]), (off[0], off[24])]
        self._workspaces = {}
        self._max_B = batch_size
        self.force_inflight = getattr(TrainEngine, "FORCE_INFLIGHT", None)   # test hook: see _set_residency (class attribute = default for new engines)
        self._use(batch_size)
        self.reducer.broadcast_(self.g_group.flat)
        self.reducer.broadcast_(self.d_group.flat)
        self.repack(G_NAMES + D_NAMES)

    @property
    def _sides(self):
        return self._sides_probed if self._use_grouped() else self._sides_legacy

    # ---- activations / workspaces: static shapes per batch size (graph-capturable), created on first use -----------
    def _use(self, B):
        """Bind the workspace set for per-GPU batch size ``B`` (the reference's DataLoader has drop_last=False, so the
        last batch of an epoch may be smaller).

        Passes that share weights and do not depend on each other run as ONE batched pass (mathematically identical:
        every op is per-sample).  Generator phase: G_A2B on [real_A|mask_A ; real_B|ones] and G_B2A on
        [real_B|mask_B ; real_A|ones] (translation + identity), then the two cycle passes.  Discriminator phase:
        each discriminator sees [real ; generated] in one pass."""
        if getattr(self, "_pending_D", None) is not None:
            self.flush()                    # a pending discriminator phase belongs to the old batch size's buffers
        if B > self._max_B:                 # a larger batch than any so far may leave the fused-trunk regime: full re-pack
            self._max_B = B
            if hasattr(self, "packed"):
                self.repack(G_NAMES + D_NAMES)         # (the discriminators' copies depend on the largest pass as well: mcvc_disc_pack_batch)
        ws = self._workspaces.get(B)
        if ws is None:
            L, T, dev = self.L, self.T, self.device
            if L.mcvc_gen_out_frames(T) != T:
                raise ValueError("training needs n_frames to be a multiple of 4 (the cycle must return the input length)")
            T8 = L.mcvc_disc_out_frames(T)
            B2, B3 = 2 * B, 3 * B
            f = lambda *s: torch.empty(s, device=dev)   # noqa: E731
            mel2 = lambda: f(B2, 80, T)                  # noqa: E731
            mel3 = lambda: f(B3, 80, T)                  # noqa: E731
            mg = self._merged_ok(B)
            ws = dict(
                g_stash2=[f(L.mcvc_gen_stash_floats(B2, T)) for _ in range(2)],      # translation+identity passes
                g_stash1=[f(L.mcvc_gen_stash_floats(B, T)) for _ in range(2)],       # cycle passes
                d_stash1=[f(L.mcvc_disc_stash_floats(B, T)) for _ in range(4)],
                d_stash2=[f(L.mcvc_disc_stash_floats(B2, T)) for _ in range(4)],
                g_scratch=[f(max(L.mcvc_gen_scratch_floats(B, T), L.mcvc_gen_scratch_floats(B2, T), L.mcvc_gen_scratch_floats(B3, T) if mg else 0))
                           for _ in range(6)],           # one per concurrent pass
                d_scratch=[f(max(L.mcvc_disc_scratch_floats(B, T), L.mcvc_disc_scratch_floats(B2, T))) for _ in range(4)],
                static_sets=[[f(B, 80, T) for _ in range(4)] for _ in range(2)],       # real_A, mask_A, real_B, mask_B (x2: pipelined step)
                g_stashD=[f(L.mcvc_gen_stash_floats(B, T)) for _ in range(2)],         # the discriminator phase's generator forwards, when pipelined
                in_A2B=mel2(), in_B2A=mel2(),                                          # [real_A ; real_B] and [real_B ; real_A]
                mask_A2B=torch.ones(B2, 80, T, device=dev), mask_B2A=torch.ones(B2, 80, T, device=dev),   # [mask ; ones]
                out_A2B=mel2(), out_B2A=mel2(),                                        # [fake_B ; identity_B], [fake_A ; identity_A]
                gout_A2B=mel2(), gout_B2A=mel2(),                                      # gradients w.r.t. those outputs
                mel={k: f(B, 80, T) for k in ("cycle_A", "cycle_B", "g_cycle_A", "g_cycle_B")},
                d_in={n: mel2() for n in D_NAMES},                                     # discriminator phase: [real ; generated]
                dout1=[f(B, 1, 10, T8) for _ in range(4)], dlogit1=[f(B, 1, 10, T8) for _ in range(4)],
                dout2=[f(B2, 1, 10, T8) for _ in range(4)], dlogit2=[f(B2, 1, 10, T8) for _ in range(4)],
            )
            if mg:
                # merged forwards (_merged_step): [identity | translation | the previous iteration's D-phase translation] per generator, and
                # [cycle | the previous iteration's D-phase cycle]
                ws.update(
                    g_stash3x=[f(L.mcvc_gen_stash_floats(B3, T)) for _ in range(2)], g_stash2c=[f(L.mcvc_gen_stash_floats(B2, T)) for _ in range(2)],
                    in3={n: mel3() for n in G_NAMES}, mask3={n: torch.ones(B3, 80, T, device=dev) for n in G_NAMES},
                    out3={n: mel3() for n in G_NAMES}, gout3={n: mel2() for n in G_NAMES},
                    cyc2={"A": mel2(), "B": mel2()},
                )
            self._workspaces[B] = ws
            for sc in ws["g_scratch"]:      # the persistent trunk kernels' error word is sticky and not initialised by the passes
                for nb in (B, B2):
                    self.L.mcvc_gen_trunk_fault(ptr(sc), nb, T, 1, stream())
        self.B = B
        for k, v in ws.items():
            setattr(self, k, v)
        self._resid = None
        self._set_residency()
        self._cur_set = 0
        self.static_in = self.static_sets[0]

    def _set_residency(self):
        """Tell the library how many persistent trunk passes this engine keeps in flight (the kernels need all their workgroups resident:
        csrc/trunk.h) -- per schedule: merged forwards = one grouped pass at a time = 2; the separate-pass pipelined schedule runs the
        D-phase's generator forwards beside the G-phase's = 4; a data-parallel rank leaves one pass's worth of compute units to RCCL's kernels.
        Re-evaluated whenever the schedule changes (the identity cut-off ends the merged schedule)."""
        if self._use_grouped():
            inflight = 4 if (not self._use_merged() and not self._serial_fwd()) else 2
        else:
            inflight = 2
        inflight += 1 if self.reducer.active else 0
        if self.force_inflight is not None:                        # (tests set this attribute to drive the residency bound into failure)
            inflight = int(self.force_inflight)
        self.L.mcvc_set_trunk_passes_in_flight(inflight)          # (process-wide in the library: re-stated every step, another engine may have changed it)
        if self._resid == (inflight, self.B):
            return
        self._resid = (inflight, self.B)
        if self._use_grouped() and not self.trunk_fallback:
            want = self.L.mcvc_gen_trunk_persistent(self.B, self.T)
            self.L.mcvc_set_trunk_passes_in_flight(1)
            could = self.L.mcvc_gen_trunk_persistent(self.B, self.T)
            self.L.mcvc_set_trunk_passes_in_flight(inflight)
            if could and not want:
                import sys
                print("[mcvc] %d persistent trunk passes in flight do not fit this device's compute units: per-layer trunk launches" % inflight,
                      file=sys.stderr, flush=True)

    # ---- thin call helpers ------------------------------------------------------------------------
    def _repack1(self, n):
        """Full refresh of one network's packed copies from its parameters: where weights arrive from outside the optimizer step
        (construction, load_state_dict, a larger batch than any so far, a phase called on its own after the caller wrote parameters)."""
        if n in G_NAMES:
            # (largest pass: two samples per input sample -- three with the merged forwards; the library sizes the copies for it)
            check(self.L.mcvc_gen_pack_ranges(self._p_tab[n], ptr(self.packed[n]), self._per_pass() * self._max_B, self.T, 3, 7, stream()), "pack " + n)
        else:
            # (every discriminator pass of this engine has at most 2 * max_B samples)
            check(self.L.mcvc_disc_pack_batch(self._p_tab[n], ptr(self.packed[n]), 2 * self._max_B, self.T, stream()), "pack " + n)

    def repack(self, names):
        """Refresh the K-major weight copies (one lane per network when running concurrently)."""
        names = list(names)
        while names:
            grp, names = names[:4], names[4:]
            self._run_tasks([(i, (lambda ln, n=n: self._repack1(n)), (), None) for i, n in enumerate(grp)])

    def _run_tasks(self, tasks):
        """Submit a phase as a dependency graph instead of fork/join rounds.

        ``tasks`` is a list of ``(lane, fn, waits, record)`` in a valid topological order: ``fn(lane)`` is queued on lane ``lane``'s HIP
        stream (lane 0 = the caller's stream) after the events named in ``waits``; ``record`` names the event recorded behind it.  Work of
        one lane is ordered by its stream, cross-lane edges are events, everything joins the caller's stream at the end.  Serial mode runs
        the same list in order on one stream."""
        if not self.concurrent or self._serial:
            for lane, fn, _, _ in tasks:
                fn(lane)
            return
        main = torch.cuda.current_stream(self.device)
        streams = [main] + self._sides
        used = sorted({t[0] for t in tasks} - {0})
        for ln in used:
            streams[ln].wait_stream(main)
        done = {}
        for ti, (lane, fn, waits, rec) in enumerate(tasks):
            st = streams[lane]
            with torch.cuda.stream(st):
                for w in waits:
                    st.wait_event(done[w])
                if self._timeline is not None:             # tools/task_timeline.py: a native (un-profiled) per-task Gantt chart
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    fn(lane)
                    e1.record(st)
                    self._timeline.append((ti, lane, "%s:%s" % (getattr(fn, "__name__", "?"), rec), waits, e0, e1))
                else:
                    fn(lane)
                if rec is not None:
                    ev = self._task_events.get(rec)
                    if ev is None:
                        ev = self._task_events[rec] = torch.cuda.Event()
                    ev.record(st)
                    done[rec] = ev
        for ln in used:
            main.wait_stream(streams[ln])

    def _aux_ptr(self, lane):
        if not self.aux_wgrad:
            return None
        if self._use_grouped():
            # grouped schedules: the generator chain's weight gradients (lane 0, the two backward rounds) run on lane 3's stream, idle by
            # then and on a hardware queue of its own
            return ctypes.c_void_p(self._sides[2].cuda_stream) if lane == 0 else None
        return ctypes.c_void_p(self._aux[lane].cuda_stream)

    def _twin(self, f0, f1):
        """Two identical call sequences on different networks as ONE set of grouped launches (gridDim.z = 2; _hip.twin)."""
        with _hip.twin() as tw:
            f0()
            tw.switch()
            f1()

    def _G(self, name, x, mask, out, stash, nb, lane=0):
        sc = self.g_scratch[lane]
        check(self.L.mcvc_gen_forward(self._p_tab[name], ptr(self.packed[name]), ptr(x), ptr(mask), ptr(out), ptr(stash), ptr(sc), sc.numel(), nb, self.T,
                                      stream()), "gen_forward")

    def _G_bwd(self, name, mask, dout, dx, acc, stash, nb, lane=0, milestones=False, aux_lane=None, ms_of=None, no_join=False, stash_nb=None, stash_b0=0):
        """``lane`` picks the scratch buffer; ``aux_lane`` (default: the same) the auxiliary weight-gradient stream -- the two halves of a
        grouped pass use different scratch buffers but the same streams and milestone events (``ms_of``).  ``stash_nb``: batch of the forward
        pass that wrote ``stash`` when this pass back-propagates through its samples [stash_b0, stash_b0 + nb) only (mcvc_gen_backward_window)."""
        sc = self.g_scratch[lane]
        ms = self._ms[ms_of or name][1] if milestones else None
        aux = self._aux_ptr(lane if aux_lane is None else aux_lane)
        check(self.L.mcvc_gen_backward_window(self._p_tab[name], ptr(self.packed[name]), self._g_tab[name], ptr(mask), ptr(dout), ptr(dx), acc, ptr(stash),
                                              stash_nb or nb, stash_b0, ptr(sc), sc.numel(), nb, self.T, stream(), aux, ms,
                                              (1 if no_join else 0) | (2 if milestones else 0)), "gen_backward")

    def _D(self, name, x, out, stash, nb, lane=0):
        sc = self.d_scratch[lane]
        check(self.L.mcvc_disc_forward(self._p_tab[name], ptr(self.packed[name]), ptr(x), ptr(out), ptr(stash), ptr(sc), sc.numel(), nb, self.T, stream()),
              "disc_forward")

    def _D_bwd(self, name, dlogit, dx, acc, stash, with_weight_grads, nb, lane=0, aux_stream=None):
        sc = self.d_scratch[lane]
        aux = ctypes.c_void_p(aux_stream.cuda_stream) if (aux_stream is not None and with_weight_grads and self.aux_wgrad) else \
            (self._aux_ptr(lane) if (with_weight_grads and self.aux_wgrad_d) else None)
        check(self.L.mcvc_disc_backward(self._p_tab[name], ptr(self.packed[name]), self._g_tab[name] if with_weight_grads else None, ptr(dlogit), 1,
                                        ptr(dx), acc, ptr(stash), ptr(sc), sc.numel(), nb, self.T, stream(), aux), "disc_backward")

    def _slot(self, i):
        return self.slots[i:i + 1]

    def _l1(self, a, b, weight, grad, k):
        """Loss call ``k`` of the iteration: value and mean go to its private pair (zeroed with the slots at the start of the iteration)."""
        check(self.L.mcvc_l1_loss(ptr(a), ptr(b), a.numel(), float(weight), ptr(self._slot(_pair(k))), ptr(self._slot(_pair(k) + 1)), ptr(grad), 0,
                                  stream()), "l1_loss")

    def _lsgan(self, d, target, weight, k, dlogit):
        check(self.L.mcvc_lsgan_loss(ptr(d), d.numel(), float(target), float(weight), ptr(self._slot(_pair(k))), ptr(self._slot(_pair(k) + 1)),
                                     ptr(dlogit), stream()), "lsgan_loss")

    def _combine(self, first, comb):
        n, loss_dst, term_dst = comb
        check(self.L.mcvc_loss_combine(ptr(self._slot(_pair(first))), n, loss_dst, term_dst, ptr(self.slots), stream()), "loss_combine")

    def _per_pass(self):
        """Samples per generator pass and input sample (the library sizes the packed copies for the largest pass)."""
        return 3 if (self.merged and self._merged_ok(self._max_B)) else 2

    def _update_gen(self, name, range_mask, lr, step, zero=True):
        """optimizer.step() on the parameter ranges ``range_mask`` of ONE generator fused with the refresh of every packed copy derived
        from them (train.py:242; mcvc_gen_update_ranges).  ``zero``: clear the gradient behind the read.  The raw-pointer update does not
        bump the parameters' autograd version counters: the module's own packed-weight caches are dropped so that a later module-API
        forward (validation, in-process inference) re-packs from the new values."""
        grp = self.g_group
        check(self.L.mcvc_gen_update_ranges(self._p_tab[name], self._numel[name], ptr(self.packed[name]), self._per_pass() * self._max_B, self.T,
                                            range_mask, ptr(grp.flat), ptr(grp.grad), None, ptr(grp.exp_avg), ptr(grp.exp_avg_sq), float(lr),
                                            self.betas[0], self.betas[1], self.eps, step, self.reducer.grad_scale, 1 if zero else 0, stream()),
              "gen_update " + name)
        self.nets[name]._packed_version = None
        self.nets[name]._bf16_version = None

    def _update_disc(self, name, lr, step, zero=True):
        """optimizer.step() of ONE discriminator fused with the refresh of its packed copies (train.py:299; mcvc_disc_update_batch)."""
        grp = self.d_group
        check(self.L.mcvc_disc_update_batch(self._p_tab[name], self._numel[name], ptr(self.packed[name]), 2 * self._max_B, self.T,
                                            ptr(grp.flat), ptr(grp.grad), None, ptr(grp.exp_avg), ptr(grp.exp_avg_sq), float(lr),
                                            self.betas[0], self.betas[1], self.eps, step, self.reducer.grad_scale, 1 if zero else 0, stream()),
              "disc_update " + name)
        self.nets[name]._packed_version = None

    def _update_discs(self, lr, zero=False):
        """All four discriminators (the plain schedules' discriminator_update): two grouped launches."""
        self.d_group.step += 1
        st = self.d_group.step
        for a, b in (D_NAMES[:2], D_NAMES[2:]):
            self._twin(lambda a=a: self._update_disc(a, lr, st, zero), lambda b=b: self._update_disc(b, lr, st, zero))

    def _update_gens(self, lr, zero=False):
        """Both generators, all ranges (the plain schedules' generator update): one grouped launch."""
        self.g_group.step += 1
        st = self.g_group.step
        self._twin(lambda: self._update_gen(G_NAMES[0], 7, lr, st, zero), lambda: self._update_gen(G_NAMES[1], 7, lr, st, zero))

    # ---- the two phases -------------------------------------------------------------------------------
    def generator_phase(self, real_A, mask_A, real_B, mask_B, fuse_update=False):
        """train.py:195-242 on four lanes (more than ``grouped_max_b`` samples per pass).  ``fuse_update`` (what ``step()`` uses): each lane
        also runs the optimizer step of the generator whose last backward pass it ran, fused with the refresh of its packed copies
        (train.py:242) -- no join of the lanes, one launch per generator."""
        B, B2 = self.B, 2 * self.B
        m = self.mel
        sc = self.sched
        self.slots[:_BLOCK].zero_()
        if self._g_grad_clean:
            self._g_grad_clean = False           # (zeroed on an idle lane during the previous discriminator phase)
        else:
            self.g_group.grad.zero_()
        # batched inputs (device-to-device copies; the batch dimension is outermost, so halves are contiguous views)
        torch._foreach_copy_([self.in_A2B[:B], self.in_A2B[B:], self.in_B2A[:B], self.in_B2A[B:], self.mask_A2B[:B], self.mask_B2A[:B]],
                             [real_A, real_B, real_B, real_A, mask_A, mask_B])     # one launch; the masks' second halves stay all-ones
        fake_B, identity_B = self.out_A2B[:B], self.out_A2B[B:]
        fake_A, identity_A = self.out_B2A[:B], self.out_B2A[B:]
        g_fake_B, g_identity_B = self.gout_A2B[:B], self.gout_A2B[B:]
        g_fake_A, g_identity_A = self.gout_B2A[:B], self.gout_B2A[B:]
        do, dl, ds = self.dout1, self.dlogit1, self.d_stash1
        # The phase as a dependency graph over four lanes.  Lane 0 follows real_A -> fake_B -> cycle_A -> D_A2 and back, lane 1 follows
        # real_B -> fake_A -> cycle_B -> D_B2 and back: these two are the critical path.  The first-step adversarial terms D_A(fake_A),
        # D_B(fake_B) need only the translated batches: lanes 2 and 3 run their forward, loss and data-gradient WHILE lanes 0/1 are in the
        # cycle forwards.  Cross-lane edges: the translated batches (g0, g1), the adversarial gradients that the cycle backward accumulates
        # onto (dA, dB), and the two backward passes of one generator, which accumulate into the same weight gradients (c0, c1).
        cl, il = sc.cycle_loss_lambda, sc.identity_loss_lambda
        # After the identity cut-off (train.py:314-315: lambda = 0 for the rest of training, 98 % of a default run) the identity passes are
        # dead code -- their loss term weighs 0 and so does every gradient behind it; like the discriminators' weight gradients of this
        # phase they are not computed: the translation passes run over B samples instead of 2B.
        nbt = B2 if il != 0 else B

        def cycle_a(ln):
            self._G("generator_B2A", fake_B, None, m["cycle_A"], self.g_stash1[0], B, ln)                  # :204 (mask of ones)
            self._l1(m["cycle_A"], real_A, cl, m["g_cycle_A"], 0)                                          # :219
            if il != 0:
                self._l1(identity_B, real_B, il, g_identity_B, 3)                                          # :224

        def cycle_b(ln):
            self._G("generator_A2B", fake_A, None, m["cycle_B"], self.g_stash1[1], B, ln)                  # :206
            self._l1(m["cycle_B"], real_B, cl, m["g_cycle_B"], 1)                                          # :220
            if il != 0:
                self._l1(identity_A, real_A, il, g_identity_A, 2)                                          # :223

        def adv(name, i, x, gx, acc):
            def run(ln):
                self._D(name, x, do[i], ds[i], B, ln)                                                      # :211-216
                self._lsgan(do[i], 1.0, 1.0, 4 + i, dl[i])                                                 # :227-231
                self._D_bwd(name, dl[i], gx, acc, ds[i], False, B, ln)         # discriminators contribute data-gradients only
            return run

        def finish_d(ln):                      # data parallel: the D gradient all-reduce of the previous iteration runs behind the
            self._finish_d_update()            # generator forwards; lane 2 is the first to need the discriminators
            if fuse_update and not self._d_grad_clean:
                # the discriminator gradients are free once their Adam step is queued: clear them here, on an idle lane, instead of
                # on the caller's stream at the top of the discriminator phase (99 MB memset)
                self.d_group.grad.zero_()
                self._d_grad_clean = True
        ov = self.overlap_g_reduce
        if fuse_update:
            self.g_group.step += 1
        g_step, g_lr = self.g_group.step, sc.g_opt_lr

        def queue_reduce(ln):
            # data parallel: the generator gradients are exchanged range by range BEHIND the last backward passes (their milestone events;
            # the tails behind the passes themselves), on the communication stream, in the same order on every rank
            for k in range(2):
                for n in G_NAMES:
                    lo, hi = self._g_ranges[n][k]
                    self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, self._ms[n][0][k])
            for n, ev in zip(G_NAMES, ("fA2B", "fB2A")):
                lo, hi = self._g_ranges[n][2]
                # serial mode records no task events (one stream: everything queued so far is the dependency); a stale event of an
                # earlier concurrent iteration must not be used either
                self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, self._task_events.get(ev) if self.concurrent else None)

        def update(name):
            def run(ln):
                self.reducer.wait(self.device)             # (no-op on one GPU)
                self._update_gen(name, 7, g_lr, g_step, zero=False)
            return run
        assert G_NAMES[0] == "generator_A2B"
        self._run_tasks([
            (0, lambda ln: self._G("generator_A2B", self.in_A2B, self.mask_A2B, self.out_A2B, self.g_stash2[0], nbt, ln), (), "g0"),   # :203, :209-210
            (1, lambda ln: self._G("generator_B2A", self.in_B2A, self.mask_B2A, self.out_B2A, self.g_stash2[1], nbt, ln), (), "g1"),   # :205, :207-208
            (0, cycle_a, (), None),
            (1, cycle_b, (), None),
            (2, finish_d, (), None),
            (2, adv("discriminator_A", 0, fake_A, g_fake_A, 0), ("g1",), "dA"),
            (3, adv("discriminator_B", 1, fake_B, g_fake_B, 0), ("g0",), "dB"),
            (0, adv("discriminator_A2", 2, m["cycle_A"], m["g_cycle_A"], 1), (), None),
            (1, adv("discriminator_B2", 3, m["cycle_B"], m["g_cycle_B"], 1), (), None),
            # backward through the cycle passes: cycle_A = G_B2A(fake_B) adds to d(fake_B), cycle_B = G_A2B(fake_A) to d(fake_A)
            (0, lambda ln: self._G_bwd("generator_B2A", None, m["g_cycle_A"], g_fake_B, 1, self.g_stash1[0], B, ln), ("dB",), "c0"),
            (1, lambda ln: self._G_bwd("generator_A2B", None, m["g_cycle_B"], g_fake_A, 1, self.g_stash1[1], B, ln), ("dA",), "c1"),
            # the last pass over each generator: its gradient ranges become final one after the other (milestone events)
            (0, lambda ln: self._G_bwd("generator_A2B", self.mask_A2B, self.gout_A2B, None, 0, self.g_stash2[0], nbt, ln, ov), ("c1",), "fA2B"),
            (1, lambda ln: self._G_bwd("generator_B2A", self.mask_B2A, self.gout_B2A, None, 0, self.g_stash2[1], nbt, ln, ov), ("c0",), "fB2A"),
        ] + ([(2, queue_reduce, (), None)] if (fuse_update and ov) else [])
          + ([(0, update("generator_A2B"), (), None), (1, update("generator_B2A"), (), None)] if fuse_update else []))
        self._g_fwd_packed = bool(fuse_update)
        self._combine(0, self._comb_g)          # g_loss and its terms, summed in the reference's order (:233-237)

    def _use_grouped(self):
        return self.grouped and self.B <= self.grouped_max_b

    # ---- grouped schedule: the two phases as parts that the plain and the pipelined step assemble into task graphs ----------------
    def _g_parts(self, inp, fuse_update, own_d_update=True, zeroing_update=False, ranged=False):
        """Closures of the generator phase (train.py:195-242) in grouped launches.  ``own_d_update``: the first-step adversarial pair
        completes a deferred discriminator update itself (plain step); the pipelined step orders the discriminators' update in front of
        the adversarial pairs through the task graph instead."""
        real_A, mask_A, real_B, mask_B = inp
        B, B2 = self.B, 2 * self.B
        m = self.mel
        sc = self.sched
        fake_B, identity_B = self.out_A2B[:B], self.out_A2B[B:]
        fake_A, identity_A = self.out_B2A[:B], self.out_B2A[B:]
        g_fake_B, g_identity_B = self.gout_A2B[:B], self.gout_A2B[B:]
        g_fake_A, g_identity_A = self.gout_B2A[:B], self.gout_B2A[B:]
        do, dl, ds = self.dout1, self.dlogit1, self.d_stash1
        cl, il = sc.cycle_loss_lambda, sc.identity_loss_lambda
        A2B, B2A = G_NAMES
        ov = self.overlap_g_reduce
        ms_on = ov or ranged                   # milestone events of the last backward pass: gradient exchange and / or ranged update
        ident_dead = il == 0                   # (after the identity cut-off the identity passes are dead code: see generator_phase)
        nbt = B if ident_dead else B2          # (the halves of the batched buffers are contiguous: translation first, identity second)
        g_lr = sc.g_opt_lr
        P = {}

        copies = ([self.in_A2B[:B], self.in_A2B[B:], self.in_B2A[:B], self.in_B2A[B:], self.mask_A2B[:B], self.mask_B2A[:B]],
                  [real_A, real_B, real_B, real_A, mask_A, mask_B])                    # (the masks' second halves stay all-ones)
        def pre(zero_grads=True):              # on the caller's stream, before the lanes fork
            self.slots[:_BLOCK].zero_()
            if zero_grads:
                if self._g_grad_clean:
                    self._g_grad_clean = False
                else:
                    self.g_group.grad.zero_()
            torch._foreach_copy_(*copies)      # one launch

        def fwd2(ln):                                                                                       # :203, :205, :207-210
            self._twin(lambda: self._G(A2B, self.in_A2B, self.mask_A2B, self.out_A2B, self.g_stash2[0], nbt, 0),
                       lambda: self._G(B2A, self.in_B2A, self.mask_B2A, self.out_B2A, self.g_stash2[1], nbt, 1))

        def cycle_half(k):
            if k == 0:
                self._G(B2A, fake_B, None, m["cycle_A"], self.g_stash1[0], B, 0)                           # :204 (mask of ones)
                self._l1(m["cycle_A"], real_A, cl, m["g_cycle_A"], 0)                                       # :219
                if not ident_dead:
                    self._l1(identity_B, real_B, il, g_identity_B, 3)                                       # :224
            else:
                self._G(A2B, fake_A, None, m["cycle_B"], self.g_stash1[1], B, 1)                           # :206
                self._l1(m["cycle_B"], real_B, cl, m["g_cycle_B"], 1)                                       # :220
                if not ident_dead:
                    self._l1(identity_A, real_A, il, g_identity_A, 2)                                       # :223

        def cycle(ln):
            self._twin(lambda: cycle_half(0), lambda: cycle_half(1))

        def adv_half(name, i, x, gx, acc):
            self._D(name, x, do[i], ds[i], B, i)                                                            # :211-216
            self._lsgan(do[i], 1.0, 1.0, 4 + i, dl[i])                                                      # :227-231
            self._D_bwd(name, dl[i], gx, acc, ds[i], False, B, i)          # discriminators contribute data-gradients only

        def adv1(ln):
            if own_d_update:
                self._finish_d_update()        # data parallel: the D all-reduce of the previous iteration hides behind the generator forwards
            self._twin(lambda: adv_half("discriminator_A", 0, fake_A, g_fake_A, 0), lambda: adv_half("discriminator_B", 1, fake_B, g_fake_B, 0))
            if own_d_update and fuse_update and not self._d_grad_clean:
                self.d_group.grad.zero_()      # free once their Adam step is queued: cleared here, on the side lane (99 MB memset)
                self._d_grad_clean = True

        def adv2(ln):
            self._twin(lambda: adv_half("discriminator_A2", 2, m["cycle_A"], m["g_cycle_A"], 1),
                       lambda: adv_half("discriminator_B2", 3, m["cycle_B"], m["g_cycle_B"], 1))

        def bwd_cycle(ln):      # cycle_A = G_B2A(fake_B) adds to d(fake_B), cycle_B = G_A2B(fake_A) to d(fake_A)
            self._twin(lambda: self._G_bwd(B2A, None, m["g_cycle_A"], g_fake_B, 1, self.g_stash1[0], B, 0, aux_lane=0),
                       lambda: self._G_bwd(A2B, None, m["g_cycle_B"], g_fake_A, 1, self.g_stash1[1], B, 1, aux_lane=0))

        def bwd_final(ln):      # the last pass over each generator; its gradient ranges become final one after the other (milestone events)
            self._twin(lambda: self._G_bwd(A2B, self.mask_A2B, self.gout_A2B, None, 0, self.g_stash2[0], nbt, 0, ms_on, aux_lane=0, ms_of=A2B),
                       lambda: self._G_bwd(B2A, self.mask_B2A, self.gout_B2A, None, 0, self.g_stash2[1], nbt, 1, ms_on, aux_lane=0, ms_of=A2B))

        def queue_reduce(ln):
            # data parallel: range k of BOTH generators is final at milestone k of the grouped pass; same collective order on every rank
            for k in range(2):
                for n in G_NAMES:
                    lo, hi = self._g_ranges[n][k]
                    self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, self._ms[A2B][0][k])
            for n in G_NAMES:
                lo, hi = self._g_ranges[n][2]
                self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, self._task_events.get("f") if self.concurrent else None)

        def update(ln):         # optimizer step of both generators + every packed copy, one grouped launch (train.py:242)
            self.reducer.wait(self.device)             # (no-op on one GPU)
            self._update_gens(g_lr, zero=zeroing_update)

        def update_range(k, last):
            """Optimizer step (+ re-pack) of part k of both generators as grouped launches: 0 = up-sampling blocks + last conv, 1 = residual
            trunk, 2 = downSample2 + conv2dto1d, 3 = downSample1, 4 = conv1 -- the order in which a backward pass finishes them.  Parts 0-3
            wait for the pass's milestone events: their gradients are final while the pass is still running, so only conv1's (0.01 % of the
            parameters) is left behind the pass."""
            mask = (1, 2, 8, 16, 32)[k]        # library range_mask of part k: [100,110), [24,100), [12,24), [4,12), [0,4)

            def run(ln):
                if k < 4:
                    torch.cuda.current_stream(self.device).wait_event(self._ms[A2B][0][k])
                if k == 0:
                    self.g_group.step += 1
                step = self.g_group.step
                self._twin(lambda: self._update_gen(A2B, mask, g_lr, step), lambda: self._update_gen(B2A, mask, g_lr, step))
            return run

        def post():
            self._g_fwd_packed = bool(fuse_update)
            self._combine(0, self._comb_g)
        P.update(update_range=update_range, pre=pre, fwd2=fwd2, cycle=cycle, adv1=adv1, adv2=adv2, bwd_cycle=bwd_cycle, bwd_final=bwd_final,
                 queue_reduce=queue_reduce, update=update, post=post, ov=ov, copies=copies)
        return P

    def generator_phase_grouped(self, real_A, mask_A, real_B, mask_B, fuse_update=False):
        """train.py:195-242 with the two generators (and each discriminator pair) in grouped launches: the same dataflow as
        ``generator_phase`` on two lanes.  Lane 0: both translation (+ identity) passes, both cycle passes, the second-step discriminators,
        both backward rounds, the update.  Lane 1: the first-step adversarial pair D_A(fake_A) | D_B(fake_B), which needs only the
        translated batches and runs beside the cycle forwards."""
        p = self._g_parts((real_A, mask_A, real_B, mask_B), fuse_update, zeroing_update=bool(fuse_update))
        p["pre"]()
        ov = p["ov"]
        self._run_tasks([
            (0, p["fwd2"], (), "g"),
            (0, p["cycle"], (), None),
            (1, p["adv1"], ("g",), "d1"),
            (0, p["adv2"], (), None),
            (0, p["bwd_cycle"], ("d1",), None),
            (0, p["bwd_final"], (), "f"),
        ] + ([(1, p["queue_reduce"], (), None)] if (fuse_update and ov) else [])
          + ([(0, p["update"], (), None)] if fuse_update else []))
        if fuse_update:
            self._g_grad_clean = True          # (cleared by the update launch that consumed them)
        p["post"]()

    def _d_parts(self, inp, gi=0, aux2=None):
        """Closures of the discriminator phase (train.py:247-299) in grouped launches.  ``gi``: index of the (stash, scratch) pair its two
        generator forwards use -- 0 = the cycle passes' (plain step: the phases run one after the other), 2 = their own (pipelined step:
        they run beside the next iteration's generator phase)."""
        real_A, mask_A, real_B, mask_B = inp
        B, B2 = self.B, 2 * self.B
        di = self.d_in
        gen_A, gen_B = di["discriminator_A"][B:], di["discriminator_B"][B:]
        cyc_A, cyc_B = di["discriminator_A2"][B:], di["discriminator_B2"][B:]
        do, dl, ds = self.dout2, self.dlogit2, self.d_stash2
        idx = {n: i for i, n in enumerate(D_NAMES)}
        A2B, B2A = G_NAMES
        gst = self.g_stash1 if gi == 0 else self.g_stashD
        s0, s1 = (0, 1) if gi == 0 else (4, 5)
        P = {}

        copies = ([di["discriminator_A"][:B], di["discriminator_A2"][:B], di["discriminator_B"][:B], di["discriminator_B2"][:B]], [real_A, real_A, real_B, real_B])
        def pre(zero_grads=True):
            self.slots[_BLOCK:].zero_()
            if zero_grads:
                if self._d_grad_clean:
                    self._d_grad_clean = False
                else:
                    self.d_group.grad.zero_()
            torch._foreach_copy_(*copies)

        def disc_full(name):
            i = idx[name]
            self._D(name, di[name], do[i], ds[i], B2, i)                       # :255-258 real half, :260-273 generated half
            self._lsgan(do[i][:B], 1.0, 0.25, 8 + 2 * i, dl[i][:B])            # every term of d_loss weighs 1/4 (:276-294)
            self._lsgan(do[i][B:], 0.0, 0.25, 9 + 2 * i, dl[i][B:])
            # (pipelined step: the second-step pair closes the critical chain; its weight gradients run on an idle lane's stream)
            self._D_bwd(name, dl[i], None, 0, ds[i], True, B2, i, aux_stream=aux2 if i >= 2 else None)

        def disc_half(name, fake):
            i = idx[name]
            sl = slice(B, B2) if fake else slice(0, B)
            st = ds[i] if fake else self.d_stash1[i]
            self._D(name, di[name][sl], do[i][sl], st, B, i)
            self._lsgan(do[i][sl], 0.0 if fake else 1.0, 0.25, 8 + 2 * i + int(fake), dl[i][sl])
            self._D_bwd(name, dl[i][sl], None, 0, st, True, B, i)

        def pair(fn, a, b, *args):
            return lambda ln: self._twin(lambda: fn(a, *args), lambda: fn(b, *args))

        def gen_fwd(ln):
            self._twin(lambda: self._G(A2B, real_A, mask_A, gen_B, gst[0], B, s0),                          # :267 generated_B
                       lambda: self._G(B2A, real_B, mask_B, gen_A, gst[1], B, s1))                          # :259 generated_A

        def cycles(ln):
            self._twin(lambda: self._G(B2A, gen_B, None, cyc_A, gst[0], B, s0),                             # :271 cycled_A
                       lambda: self._G(A2B, gen_A, None, cyc_B, gst[1], B, s1))                             # :263 cycled_B

        def repack_full(ln):
            self._twin(lambda: self._repack1(A2B), lambda: self._repack1(B2A))

        def post():
            self._combine(8, self._comb_d)
        P.update(pre=pre, gen_fwd=gen_fwd, cycles=cycles, repack_full=repack_full, post=post, copies=copies,
                 full1=pair(disc_full, "discriminator_A", "discriminator_B"), full2=pair(disc_full, "discriminator_A2", "discriminator_B2"),
                 real1=pair(disc_half, "discriminator_A", "discriminator_B", False), real2=pair(disc_half, "discriminator_A2", "discriminator_B2", False),
                 fake1=pair(disc_half, "discriminator_A", "discriminator_B", True), fake2=pair(disc_half, "discriminator_A2", "discriminator_B2", True))
        return P

    def discriminator_phase_grouped(self, real_A, mask_A, real_B, mask_B):
        """train.py:247-299, grouped like ``generator_phase_grouped``: lane 0 runs both generators' forwards (translation, then cycle) and
        the second-step discriminator pair, lane 1 runs D_A | D_B."""
        p = self._d_parts((real_A, mask_A, real_B, mask_B))
        p["pre"]()
        packed = self._g_fwd_packed
        self._g_fwd_packed = False

        def gen_fwd(ln):
            if not packed:                     # phase called on its own (the caller may have written parameters): full refresh first
                p["repack_full"](ln)
            p["gen_fwd"](ln)

        def refresh(ln):                       # beside the generator forwards; the generator gradients are free by now
            if packed and not self._g_grad_clean:
                self.g_group.grad.zero_()      # (196 MB memset on the side lane instead of at the top of the next iteration)
                self._g_grad_clean = True
        if self.B >= self.split_d_min_batch:
            # the real halves need nothing from the generators: lane 1 runs them while lane 0 is in the generator forwards
            tasks = [
                (0, gen_fwd, (), "gen"),
                (1, refresh, (), None),
                (1, p["real1"], (), None),
                (1, p["real2"], (), "r2"),
                (0, p["cycles"], (), None),
                (1, p["fake1"], ("gen",), None),
                (0, p["fake2"], ("r2",), None),
            ]
        else:
            tasks = [
                (0, gen_fwd, (), "gen"),
                (1, refresh, (), None),
                (0, p["cycles"], (), None),
                (1, p["full1"], ("gen",), None),
                (0, p["full2"], (), None),
            ]
        self._run_tasks(tasks)
        p["post"]()

    def _d_pair_update(self, pair, d_lr, d_step):
        """Task: optimizer step of one discriminator pair (its slice of the flat buffer; data parallel: behind its all-reduce) fused with the
        refresh of its packed copies -- one grouped launch (train.py:299)."""
        lo, hi = self._d_ranges[pair[0]][0], self._d_ranges[pair[1]][1]

        def run(ln):
            if self.reducer.active:
                self.reducer.reduce_range_after_(self.d_group.grad, lo, hi, None)
                self.reducer.wait(self.device)
            self._twin(lambda: self._update_disc(pair[0], d_lr, d_step), lambda: self._update_disc(pair[1], d_lr, d_step))
        return run

    # ---- pipelined step ------------------------------------------------------------------------------------------------------------
    def _pipelined_step(self):
        """Iteration t+1's generator phase beside iteration t's discriminator phase.

        The discriminator phase of an iteration (train.py:247-299) reads the generators as updated by that iteration's generator phase
        and writes the discriminators; the NEXT iteration's generator forwards (translation, identity, cycle: train.py:203-210) read the
        same generator weights and no discriminator at all -- only its adversarial passes (:211-216) need the updated discriminators,
        and each pair only its own two networks.  So a step issues, as ONE task graph:
            lane 1:  D-phase(t): generator forwards -> cycle forwards -> D_A2 | D_B2 -> [all-reduce] update of their slice -> "dupd2";
                     then the ranged generator update of G-phase(t+1) behind the last backward pass's milestone events
            lane 3:  D_A | D_B of D-phase(t) -> update of their slice -> "dupd1"; the backward rounds' weight gradients
            lane 0:  G-phase(t+1): translation (+ identity) -> cycle -> (dupd2) D_A2 | D_B2 adversarial -> backward x 2 -> conv1's update
            lane 2:  (dupd1) the first-step adversarial pair of G-phase(t+1)
        Every quantity is computed from exactly the weights and inputs the reference uses (the order of the optimizers' steps is kept:
        a discriminator's Adam step (t) is complete before it is read by iteration t+1; Adam(G)(t+1) waits for D-phase(t)'s generator
        forwards); two generator-forward latencies leave the critical path of every iteration.  The update launches clear the gradients
        they consume: no per-iteration gradient memsets.  ``d_loss`` of an iteration becomes available one ``step()`` later
        (``losses(lagged=True)``); ``flush()`` / ``losses()`` complete a pending phase."""
        cur = self.static_in
        prev, d_lr = self._pending_D if self._pending_D is not None else (None, None)
        ranged = not self.reducer.active
        g = self._g_parts(cur, True, own_d_update=prev is None, zeroing_update=True, ranged=ranged)
        ov = g["ov"]
        if prev is None:                       # first iteration (or the first after a flush): there is no discriminator phase to run beside it
            g["pre"](zero_grads=True)
            tasks = [(0, g["fwd2"], (), "g"), (0, g["cycle"], (), None), (2, g["adv1"], ("g",), "d1"), (0, g["adv2"], (), None),
                     (0, g["bwd_cycle"], ("d1",), None), (0, g["bwd_final"], (), "f")]
            tasks += ([(3, g["queue_reduce"], (), None)] if ov else []) + [(0, g["update"], (), None)]
            self._run_tasks(tasks)
            self._g_grad_clean = True          # (cleared by the update launch that consumed them)
            g["post"]()
            return
        d = self._d_parts(prev, gi=2, aux2=self._sides[1])            # (lane 2's stream: idle until "dupd1")
        # caller's stream, before the lanes fork, TWO launches (five until r5, ~15 us of idle GPU each at the head of the chain): iteration t's g_loss
        # terms saved + both phases' input copies; both loss blocks cleared (the gradients were cleared by the updates that consumed them)
        torch._foreach_copy_([self.slots_done[:_BLOCK]] + g["copies"][0] + d["copies"][0], [self.slots[:_BLOCK]] + g["copies"][1] + d["copies"][1])
        self.slots.zero_()
        packed = self._g_fwd_packed
        d_step = self.d_group.step + 1
        split, tail = self.B >= self.split_d_min_batch, self.concurrent and not self._serial
        serial = self._serial_fwd()            # (data-parallel ranks: one grouped persistent trunk pass in flight at a time)
        tasks = [
            (1, (lambda ln: (None if packed else d["repack_full"](ln), d["gen_fwd"](ln))), (), "gen"),
        ] + ([(1, d["cycles"], (), "cyc")] if serial else []) + [
            (0, g["fwd2"], ("cyc",) if serial else (), "g"),
        ]
        if split:
            tasks += [(3, d["real1"], (), None), (3, d["real2"], (), "r2")]
        tasks += ([] if serial else [(1, d["cycles"], (), "cyc")]) + [
            (0, g["cycle"], (), None),
            (3, d["fake1"] if split else d["full1"], ("gen",), None),
            (3, self._d_pair_update(("discriminator_A", "discriminator_B"), d_lr, d_step), (), "dupd1"),
            (1, d["fake2"] if split else d["full2"], ("r2",) if split else (), None),
            (1, self._d_pair_update(("discriminator_A2", "discriminator_B2"), d_lr, d_step), (), "dupd2"),
            (2, g["adv1"], ("g", "dupd1"), "d1"),
        ] + ([(2, lambda ln: (d["post"](), self.slots_done[_BLOCK:].copy_(self.slots[_BLOCK:]), self._publish_done()), ("dupd2",), None)] if tail else []) + [
            (0, g["adv2"], ("dupd2",), None),
            (0, g["bwd_cycle"], ("d1",), None),
            (0, g["bwd_final"], (), "f"),
        ] + ([(2, lambda ln: g["post"](), ("f",), None)] if tail else []) + ([(3, g["queue_reduce"], (), None)] if ov else [])
        if ranged:
            # the generator update range by range: the up-sampling blocks', the trunk's and most of the head's optimizer step + re-pack run
            # on lane 1 (idle since "dupd2" -- lane 3's stream carries the backward rounds' weight gradients) beside the rest of the last
            # backward pass, behind its milestone events and after everything that still reads the old weights (D-phase(t)'s generator
            # forwards: "cyc"); only conv1's is left for the end of the chain
            tasks += [(1, g["update_range"](0, False), ("cyc",), None), (1, g["update_range"](1, False), (), None),
                      (1, g["update_range"](2, False), (), None), (1, g["update_range"](3, False), (), None),
                      (0, g["update_range"](4, True), ("cyc",), None)]
        else:
            tasks += [(0, g["update"], ("cyc",), None)]
        self._run_tasks(tasks)
        self.d_group.step = d_step
        self._g_grad_clean = self._d_grad_clean = True
        if not tail:      # (tail: the loss sums and the publish of iteration t's losses ride on lanes 1 / 2 instead of the end of the chain)
            g["post"](), d["post"](), self.slots_done[_BLOCK:].copy_(self.slots[_BLOCK:]), self._publish_done()

    # ---- merged forwards (r4) ---------------------------------------------------------------------------------------------------------
    def _merged_ok(self, B):
        """Three samples per generator pass must stay inside the fused 1-D trunk kernels (csrc/trunk.h: 48 columns at 64 frames)."""
        return 3 * B * (self.T // 4) <= 48 and self.T % 4 == 0

    def _serial_fwd(self):
        """Data-parallel ranks on the separate-pass pipelined schedule (the merged schedule ends with the identity cut-off: 98 % of a default
        run) order the G-phase's grouped forwards BEHIND the D-phase's generator forwards, so that ONE grouped persistent trunk pass is in
        flight at a time, like on the merged schedule: 2 + 1 (RCCL's share) x 64 workgroups fit the 256 compute units, where the free-running
        4 + 1 do not and every pass would fall back to per-layer trunk launches.  The D-phase chain is the critical one; the
        G-phase forwards have ~0.7 ms of slack behind it."""
        return self.reducer.active and self._use_pipeline() and not self._use_merged()

    def _use_merged(self):
        # (after the identity cut-off the merged passes would carry a dead sample: the separate passes skip it)
        return self.merged and self._use_pipeline() and self._merged_ok(self.B) and hasattr(self, "in3") and self.sched.identity_loss_lambda != 0

    def _merged_step(self):
        """``_pipelined_step`` with the discriminator phase's generator forwards INSIDE the generator phase's passes.

        D-phase(t) needs generated = G(real(t), mask(t)) and cycled = G'(generated) with the weights Adam(G)(t) left (train.py:259-273), and
        discards their backward (reset_grad at :240 of t+1).  G-phase(t+1) runs translation + identity and then cycle forwards with exactly
        those weights (:203-210).  So one grouped pass per generator carries THREE samples
            [ identity: real_other(t+1) | ones ;  translation: real(t+1) | mask(t+1) ;  D-phase translation: real(t) | mask(t) ]
        and the cycle pass TWO: [ fake(t+1) ; generated(t) ] (contiguous rows of the first pass's output).  The backward passes run over the
        first two / the first sample of those stashes (mcvc_gen_backward_window).  Per-sample results are what the separate passes compute
        (every op of the generator is per sample): tests/test_hip_twin.py.  Two of six grouped generator passes disappear; one persistent trunk pass in flight, not two.
            lane 0: forward x3 -> cycle x2 -> the two backward passes -> conv1's update;  lanes 1 / 2: D_A | D_B (D_A2 | D_B2) of D-phase(t) ->
            their update -> first- (second-) step adversarial pair of G-phase(t+1), lane 1 then the ranged generator update;  lane 3: weight gradients"""
        cur = self.static_in
        prev, d_lr = self._pending_D if self._pending_D is not None else (None, None)
        B, B2, B3 = self.B, 2 * self.B, 3 * self.B
        A2B, B2A = G_NAMES
        sc = self.sched
        cl, il = sc.cycle_loss_lambda, sc.identity_loss_lambda
        ranged = not self.reducer.active
        g = self._g_parts(cur, True, own_d_update=False, zeroing_update=True, ranged=ranged)     # (update / reduce / post closures)
        ov = g["ov"]
        ms_on = ov or ranged
        real_A, mask_A, real_B, mask_B = cur
        p_real_A, p_mask_A, p_real_B, p_mask_B = prev if prev is not None else cur          # (first iteration: a throw-away third sample)
        in3, mask3, out3, gout3, cyc2 = self.in3, self.mask3, self.out3, self.gout3, self.cyc2
        identity_B, fake_B, gen_B = out3[A2B][:B], out3[A2B][B:B2], out3[A2B][B2:]
        identity_A, fake_A, gen_A = out3[B2A][:B], out3[B2A][B:B2], out3[B2A][B2:]
        g_identity_B, g_fake_B = gout3[A2B][:B], gout3[A2B][B:]
        g_identity_A, g_fake_A = gout3[B2A][:B], gout3[B2A][B:]
        cycle_A, cyc_A = cyc2["A"][:B], cyc2["A"][B:]
        cycle_B, cyc_B = cyc2["B"][:B], cyc2["B"][B:]
        m = self.mel
        di = self.d_in
        do, dl, ds = self.dout1, self.dlogit1, self.d_stash1
        d = self._d_parts(prev, gi=2, aux2=None) if prev is not None else None
        if prev is not None:
            self.slots_done[:_BLOCK].copy_(self.slots[:_BLOCK])       # g_loss and its terms of iteration t, before the block is reused
        # ---- on the caller's stream, before the lanes fork
        self.slots[:_BLOCK].zero_()
        if not self._g_grad_clean:
            self.g_group.grad.zero_()
        torch._foreach_copy_(
            [in3[A2B][:B], in3[A2B][B:B2], in3[A2B][B2:], mask3[A2B][B:B2], mask3[A2B][B2:],
             in3[B2A][:B], in3[B2A][B:B2], in3[B2A][B2:], mask3[B2A][B:B2], mask3[B2A][B2:]],
            [real_B, real_A, p_real_A, mask_A, p_mask_A,
             real_A, real_B, p_real_B, mask_B, p_mask_B])                # (the identity samples' masks stay all-ones)
        if d is not None:
            d["pre"](zero_grads=not self._d_grad_clean)
        packed = self._g_fwd_packed
        d_step = self.d_group.step + 1

        def fwd3(ln):                                                                                       # :203, :205, :207-210 | :259, :267
            if not packed:
                self._twin(lambda: self._repack1(A2B), lambda: self._repack1(B2A))
            self._twin(lambda: self._G(A2B, in3[A2B], mask3[A2B], out3[A2B], self.g_stash3x[0], B3, 0),
                       lambda: self._G(B2A, in3[B2A], mask3[B2A], out3[B2A], self.g_stash3x[1], B3, 1))
            if d is not None:
                torch._foreach_copy_([di["discriminator_B"][B:], di["discriminator_A"][B:]], [gen_B, gen_A])

        def cycle2(ln):                                                                                     # :204, :206 | :263, :271
            self._twin(lambda: self._G(B2A, out3[A2B][B:], None, cyc2["A"], self.g_stash2c[0], B2, 0),
                       lambda: self._G(A2B, out3[B2A][B:], None, cyc2["B"], self.g_stash2c[1], B2, 1))
            if d is not None:
                torch._foreach_copy_([di["discriminator_A2"][B:], di["discriminator_B2"][B:]], [cyc_A, cyc_B])
            self._twin(lambda: (self._l1(cycle_A, real_A, cl, m["g_cycle_A"], 0), self._l1(identity_B, real_B, il, g_identity_B, 3)),   # :219, :224
                       lambda: (self._l1(cycle_B, real_B, cl, m["g_cycle_B"], 1), self._l1(identity_A, real_A, il, g_identity_A, 2)))   # :220, :223

        def adv_half(name, i, x, gx, acc):
            self._D(name, x, do[i], ds[i], B, i)                                                            # :211-216
            self._lsgan(do[i], 1.0, 1.0, 4 + i, dl[i])                                                      # :227-231
            self._D_bwd(name, dl[i], gx, acc, ds[i], False, B, i)          # discriminators contribute data-gradients only

        def adv1(ln):
            self._twin(lambda: adv_half("discriminator_A", 0, fake_A, g_fake_A, 0), lambda: adv_half("discriminator_B", 1, fake_B, g_fake_B, 0))

        def adv2(ln):
            self._twin(lambda: adv_half("discriminator_A2", 2, cycle_A, m["g_cycle_A"], 1),
                       lambda: adv_half("discriminator_B2", 3, cycle_B, m["g_cycle_B"], 1))

        # The cycle pass's weight gradients (auxiliary stream) outlast its data-gradient chain by ~0.2 ms; the translation pass needs only the
        # data gradient, so it starts without that join (MCVC_BWD_NO_JOIN, include/mcvc.h): its own weight gradients queue behind them on the
        # same auxiliary stream (same tensors, in order), it works in other scratch buffers (4, 5), and its final join covers both passes.
        def bwd_cycle(ln):      # cycle_A = G_B2A(fake_B) adds to d(fake_B), cycle_B = G_A2B(fake_A) to d(fake_A)
            self._twin(lambda: self._G_bwd(B2A, None, m["g_cycle_A"], g_fake_B, 1, self.g_stash2c[0], B, 0, aux_lane=0, no_join=True, stash_nb=B2),
                       lambda: self._G_bwd(A2B, None, m["g_cycle_B"], g_fake_A, 1, self.g_stash2c[1], B, 1, aux_lane=0, no_join=True, stash_nb=B2))

        def bwd_final(ln):      # identity + translation samples; the gradient ranges become final one after the other (milestone events)
            self._twin(lambda: self._G_bwd(A2B, mask3[A2B], gout3[A2B], None, 0, self.g_stash3x[0], B2, 4, ms_on, aux_lane=0, ms_of=A2B, stash_nb=B3),
                       lambda: self._G_bwd(B2A, mask3[B2A], gout3[B2A], None, 0, self.g_stash3x[1], B2, 5, ms_on, aux_lane=0, ms_of=A2B, stash_nb=B3))

        tasks = [(0, fwd3, (), "g"), (0, cycle2, (), "c")]
        if d is not None:
            tasks += [(1, d["full1"], ("g",), None), (1, self._d_pair_update(("discriminator_A", "discriminator_B"), d_lr, d_step), (), None),
                      (2, d["full2"], ("c",), None), (2, self._d_pair_update(("discriminator_A2", "discriminator_B2"), d_lr, d_step), (), None)]
        tasks += [(1, adv1, ("g",), "d1"), (2, adv2, ("c",), "d2"),
                  (0, bwd_cycle, ("d1", "d2"), None), (0, bwd_final, (), "f")]
        if ov:
            tasks += [(3, g["queue_reduce"], (), None)]
        if ranged:
            tasks += [(1, g["update_range"](0, False), (), None), (1, g["update_range"](1, False), (), None),
                      (1, g["update_range"](2, False), (), None), (1, g["update_range"](3, False), (), None), (0, g["update_range"](4, True), (), None)]
        else:
            tasks += [(0, g["update"], (), None)]
        self._run_tasks(tasks)
        self._g_grad_clean = True              # (cleared by the update launches that consumed them)
        g["post"]()
        if d is not None:
            self.d_group.step = d_step
            self._d_grad_clean = True
            d["post"]()
            self.slots_done[_BLOCK:].copy_(self.slots[_BLOCK:])           # d_loss and its terms of iteration t: the iteration is complete
            self._publish_done()

    def _publish_done(self):
        """Losses of the last COMPLETE iteration to pinned host memory (asynchronous copy + event; ``losses(lagged=True)`` waits for it)."""
        host, ev = self._done_ring[self._done_count % 4]
        host.copy_(self.slots_done, non_blocking=True)
        ev.record()
        self._done_count += 1

    def generator_update(self):
        """All-reduce (data parallel) + optimizer step of both generators (train.py:242) behind a generator phase that ran without its own."""
        if self.overlap_g_reduce:
            # same collective order on every rank: range k of A2B, range k of B2A, k = 0, 1, then the two tails
            for k in range(2):
                for n in G_NAMES:
                    lo, hi = self._g_ranges[n][k]
                    self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, self._ms[n][0][k])
            for n in G_NAMES:
                lo, hi = self._g_ranges[n][2]
                self.reducer.reduce_range_after_(self.g_group.grad, lo, hi, None)
            self.reducer.wait(self.device)
        else:
            self.reducer.reduce_(self.g_group.grad)
        self._update_gens(self.sched.g_opt_lr)

    def discriminator_phase(self, real_A, mask_A, real_B, mask_B):
        """train.py:247-299 on four lanes."""
        B, B2 = self.B, 2 * self.B
        self.slots[_BLOCK:].zero_()
        if self._d_grad_clean:
            self._d_grad_clean = False
        else:
            self.d_group.grad.zero_()
        di = self.d_in
        # generators run with their UPDATED weights and no gradient (train.py:259-273); outputs land directly in the
        # second half of the discriminators' batched inputs
        gen_A, gen_B = di["discriminator_A"][B:], di["discriminator_B"][B:]
        cyc_A, cyc_B = di["discriminator_A2"][B:], di["discriminator_B2"][B:]
        torch._foreach_copy_([di["discriminator_A"][:B], di["discriminator_A2"][:B], di["discriminator_B"][:B], di["discriminator_B2"][:B]],
                             [real_A, real_A, real_B, real_B])
        do, dl, ds = self.dout2, self.dlogit2, self.d_stash2
        idx = {n: i for i, n in enumerate(D_NAMES)}

        def disc(name):
            i = idx[name]

            def run(ln):
                self._D(name, di[name], do[i], ds[i], B2, ln)                      # :255-258 real half, :260-273 generated half
                # d_loss = (A + B)/2 + (A_2nd + B_2nd)/2 with each = (real + fake)/2  -> every term weighs 1/4  (:276-294)
                self._lsgan(do[i][:B], 1.0, 0.25, 8 + 2 * i, dl[i][:B])
                self._lsgan(do[i][B:], 0.0, 0.25, 9 + 2 * i, dl[i][B:])
                self._D_bwd(name, dl[i], None, 0, ds[i], True, B2, ln)
            return run
        # Same shape as the generator phase: lanes 0/1 carry the two generator chains, D_A / D_B need only the generated batches and run
        # on lanes 2/3 while lanes 0/1 are in the cycle forwards; the second-step discriminators follow the cycle forwards on lanes 0/1.
        if self._g_fwd_packed:
            # the generator phase left lane 0 with generator_A2B updated and its copies fresh, lane 1 with generator_B2A: each lane starts
            # with its own generator
            self._g_fwd_packed = False

            def clear_g(ln):
                self.g_group.grad.zero_()      # the generator gradients are free (their update ended the generator phase): 196 MB
                self._g_grad_clean = True      # memset on an idle lane instead of at the top of the next iteration
            head = [
                (0, lambda ln: self._G("generator_A2B", real_A, mask_A, gen_B, self.g_stash1[0], B, ln), (), "gB"),      # :267 generated_B
                (1, lambda ln: self._G("generator_B2A", real_B, mask_B, gen_A, self.g_stash1[1], B, ln), (), "gA"),      # :259 generated_A
                (2, clear_g, (), None),
            ]
            cycles = [
                (0, lambda ln: self._G("generator_B2A", gen_B, None, cyc_A, self.g_stash1[0], B, ln), (), None),         # :271 cycled_A
                (1, lambda ln: self._G("generator_A2B", gen_A, None, cyc_B, self.g_stash1[1], B, ln), (), None),         # :263 cycled_B
            ]
            if B >= self.split_d_min_batch:
                # The real halves need nothing from the generators: lanes 2/3 run them while lanes 0/1 are in the generator forwards, so the
                # tail of the phase (second-step discriminators behind the cycle forwards) is a pass over the generated half only.  A
                # discriminator's two passes add into the same gradients: same lane, or ordered by an event.
                def half(name, fake):
                    i = idx[name]
                    sl = slice(B, B2) if fake else slice(0, B)
                    st = ds[i] if fake else self.d_stash1[i]

                    def run(ln):
                        self._D(name, di[name][sl], do[i][sl], st, B, ln)
                        self._lsgan(do[i][sl], 0.0 if fake else 1.0, 0.25, 8 + 2 * i + int(fake), dl[i][sl])
                        self._D_bwd(name, dl[i][sl], None, 0, st, True, B, ln)
                    return run
                tasks = head + [
                    (2, half("discriminator_A", False), (), None),
                    (3, half("discriminator_B", False), (), None),
                    (2, half("discriminator_A2", False), (), "rA2"),
                    (3, half("discriminator_B2", False), (), "rB2"),
                ] + cycles + [
                    (2, half("discriminator_A", True), ("gA",), None),
                    (3, half("discriminator_B", True), ("gB",), None),
                    (0, half("discriminator_A2", True), ("rA2",), None),
                    (1, half("discriminator_B2", True), ("rB2",), None),
                ]
            else:
                tasks = head + cycles + [
                    (2, disc("discriminator_A"), ("gA",), None),
                    (3, disc("discriminator_B"), ("gB",), None),
                    (0, disc("discriminator_A2"), (), None),
                    (1, disc("discriminator_B2"), (), None),
                ]
        else:
            # (phase called on its own: a lane re-packs the generator it runs first; the other lane's second pass waits for that re-pack)
            tasks = [
                (0, lambda ln: self._repack1("generator_B2A"), (), "p0"),
                (1, lambda ln: self._repack1("generator_A2B"), (), "p1"),
                (0, lambda ln: self._G("generator_B2A", real_B, mask_B, gen_A, self.g_stash1[0], B, ln), (), "gA"),      # :259 generated_A
                (1, lambda ln: self._G("generator_A2B", real_A, mask_A, gen_B, self.g_stash1[1], B, ln), (), "gB"),      # :267 generated_B
                (0, lambda ln: self._G("generator_A2B", gen_A, None, cyc_B, self.g_stash1[0], B, ln), ("p1",), None),    # :263 cycled_B
                (1, lambda ln: self._G("generator_B2A", gen_B, None, cyc_A, self.g_stash1[1], B, ln), ("p0",), None),    # :271 cycled_A
                (2, disc("discriminator_A"), ("gA",), None),
                (3, disc("discriminator_B"), ("gB",), None),
                (0, disc("discriminator_B2"), (), None),
                (1, disc("discriminator_A2"), (), None),
            ]
        self._run_tasks(tasks)
        self._combine(8, self._comb_d)

    def discriminator_update(self):
        """train.py:299.  With more than one rank the all-reduce is only *started* here; the optimizer step runs when the discriminators
        are next needed (``_finish_d_update``), so the exchange overlaps the next iteration's generator forwards."""
        if self.defer_d_update:
            self.reducer.reduce_async_(self.d_group.grad)
            self._pending_d_lr = self.sched.d_opt_lr          # the value torch.optim would have used now
            return
        self.reducer.reduce_(self.d_group.grad)
        self._update_discs(self.sched.d_opt_lr)

    def _finish_d_update(self):
        if self._pending_d_lr is None:
            return
        self.reducer.wait(self.device)
        lr, self._pending_d_lr = self._pending_d_lr, None
        self._update_discs(lr)

    def _use_pipeline(self):
        return self.pipelined and self._use_grouped() and self.concurrent and (not self.reducer.active or self.overlap_g_reduce)

    def _next_input_set(self):
        """The static input buffers the coming iteration writes its minibatch into: the other set while the previous iteration's
        discriminator phase (which still reads its own minibatch) is pending."""
        if self._pending_D is not None:
            self._cur_set ^= 1
            self.static_in = self.static_sets[self._cur_set]

    def flush(self):
        """Complete what ``step()`` left pending -- the pipelined discriminator phase of the last iteration, a deferred discriminator
        update -- so that parameters, optimizer state and both losses belong to the same, complete iteration."""
        if self._pending_D is not None:
            inp, d_lr = self._pending_D
            self._pending_D = None
            self.slots_done[:_BLOCK].copy_(self.slots[:_BLOCK])
            self.discriminator_phase_grouped(*inp)
            self.reducer.reduce_(self.d_group.grad)
            self._update_discs(d_lr, zero=True)
            self._d_grad_clean = True
            self.slots_done[_BLOCK:].copy_(self.slots[_BLOCK:])
            self._publish_done()
        self._finish_d_update()

    def step(self, real_A, mask_A, real_B, mask_B):
        """One full iteration.  Inputs: float32 [B,80,T] on the engine's device.  Returns the loss-slot
        tensor (device); ``losses()`` does the host read the reference does with ``.item()`` (train.py:303).
        With more than one rank the discriminators' Adam step of this iteration is queued by the next ``step()`` (or by ``flush()``):
        call ``flush()`` before reading parameters or optimizer state from outside the engine."""
        _hip.require_cuda_f32(real_A, mask_A, real_B, mask_B)
        if tuple(real_A.shape[1:]) != (80, self.T):
            raise ValueError("batch shape %s does not match the engine (B, 80, %d)" % (tuple(real_A.shape), self.T))
        if real_A.shape[0] != self.B:
            self._use(int(real_A.shape[0]))
        self._next_input_set()
        torch._foreach_copy_(list(self.static_in), [real_A, mask_A, real_B, mask_B])
        return self._step_static()

    def step_sampled(self, sampler, batch_size=None):
        """One full iteration on a minibatch drawn ON THE DEVICE (dataset.device_sampler.DeviceSampler): the sampler's kernel
        writes crops and masks straight into the static input buffers -- no DataLoader, no H2D copy."""
        B = self.B if batch_size is None else int(batch_size)
        if B != self.B:
            self._use(B)
        self._next_input_set()
        sampler.draw_into(*self.static_in)
        return self._step_static()

    def _step_static(self):
        self._set_residency()
        if self._use_pipeline():
            d_lr = self.sched.d_opt_lr           # what torch.optim would use for THIS iteration's discriminator step
            if self._use_merged():
                self._merged_step()              # ... with the previous discriminator phase's generator forwards inside its own passes
            else:
                self._pipelined_step()           # generator phase of this iteration (+ the previous one's discriminator phase)
            self._pending_D = (self.static_in, d_lr)
            self.sched.end_iteration()
            return self.slots
        if self._pending_D is not None:
            self.flush()
        if not self.reducer.active or self.overlap_g_reduce:
            phase = self.generator_phase_grouped if self._use_grouped() else self.generator_phase
            phase(*self.static_in, fuse_update=True)                      # includes the generator update (no join of the lanes)
        else:
            self._run_phase("G")
            self.generator_update()
        self._run_phase("D")
        self.discriminator_update()
        self.sched.end_iteration()
        return self.slots

    def _run_phase(self, which):
        if self._use_grouped():
            fn = self.generator_phase_grouped if which == "G" else self.discriminator_phase_grouped
        else:
            fn = self.generator_phase if which == "G" else self.discriminator_phase
        fn(*self.static_in)

    def check_faults(self, raise_on_fault=True):
        """Detect a persistent trunk launch of this engine that gave up waiting for its workgroups (its result was poisoned with NaN, so
        the losses are non-finite as well).  On a fault: the error words are cleared, the library is switched to per-layer trunk launches
        for the rest of the process (the condition that lost an arrival -- more persistent passes resident than the device holds, or
        foreign kernels occupying compute units for longer than the bounded spin -- would recur), the switch is logged, and a
        RuntimeError names the layer (``raise_on_fault=False``: the code is returned instead; the caller restores parameters and optimizer
        state from before the poisoned step and continues).  Synchronises the device: call it per logging interval, not per step."""
        first = 0
        for B, ws in self._workspaces.items():
            for lane, sc in enumerate(ws["g_scratch"]):
                code = self.L.mcvc_gen_trunk_fault(ptr(sc), B, self.T, 1, stream())          # (the word sits at the same place for every batch)
                if code != 0 and not first:
                    first = (code, lane, B)
        if not first:
            return 0
        code, lane, B = first
        self.trunk_fallback = True
        self.L.mcvc_set_trunk_persistent(0)
        import sys
        print("[mcvc] persistent trunk kernel fault %d (layer %d) in lane %d: switching to per-layer trunk launches" % (code, (code & 0xff) - 1, lane),
              file=sys.stderr, flush=True)
        if raise_on_fault:
            raise RuntimeError("persistent trunk kernel fault %d (layer %d) in lane %d at batch %d: more concurrent generator passes than the "
                               "device can keep resident; the step's result is poisoned -- restore the last good state (per-layer launches "
                               "are now in effect)" % (code, (code & 0xff) - 1, lane, B))
        return code

    def losses(self, lagged=False):
        """Host read of the loss slots, like the reference's ``.item()`` calls (train.py:303).  Default: the losses of the iteration
        just issued -- a pending pipelined discriminator phase is completed first, so calling this every iteration runs the two phases
        back to back.  ``lagged=n`` (True = 1; the pipelined training loop and bench.py use 2): the n-th newest COMPLETE iteration -- after
        ``step()`` number t+1 that is iteration t+1-n (None before there is one) -- waiting only for the asynchronous copy that published
        it.  n = 1 still waits for the discriminator phase queued by the last ``step()``, i.e. for the GPU to drain; n = 2 never waits."""
        if lagged:
            k = self._done_count - max(1, int(lagged))
            if self._pending_D is None:
                lagged = False                   # nothing in flight: the latest iteration is complete
            elif k < self._done_base:
                return None
            else:
                host, ev = self._done_ring[k % 4]
                ev.synchronize()
                v = host.tolist()
                return {"g_loss": v[SLOT_G], "d_loss": v[SLOT_D], "cycle_loss": v[SLOT_CYCLE], "identity_loss": v[SLOT_IDENT],
                        "adv_loss": v[SLOT_ADV_G]}
        if self._pending_D is not None:
            self.flush()
        self._done_base = self._done_count      # (what was published so far is older than what this read returns)
        v = self.slots.tolist()      # device sync, like the reference's .item()
        return {"g_loss": v[SLOT_G], "d_loss": v[SLOT_D], "cycle_loss": v[SLOT_CYCLE], "identity_loss": v[SLOT_IDENT],
                "adv_loss": v[SLOT_ADV_G]}

    def optimizer(self, which):
        """Adapter with ``state_dict()`` / ``load_state_dict()`` in torch.optim.Adam layout (for saver.ModelSaver)."""
        eng = self

        class _Adapter(object):
            def state_dict(self):
                return eng.optimizer_state_dict(which)

            def load_state_dict(self, sd):
                eng.load_optimizer_state_dict(which, sd)
        return _Adapter()

    # ---- torch.optim.Adam-compatible optimizer state (checkpoint layout of the reference): optim_state.py
    def optimizer_state_dict(self, which):
        return optimizer_state_dict(self, which)

    def load_optimizer_state_dict(self, which, sd):
        load_optimizer_state_dict(self, which, sd)
