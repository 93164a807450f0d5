/* mcvc.h -- C ABI of the MI355X-native MaskCycleGAN-VC hot path (libmcvc_hip.so, gfx950 only).
 *
 * The reference (GANtastic3/MaskCycleGAN-VC) has no FFI: its hot path is Python calling torch.nn
 * ops.  This ABI is the drop-in boundary *below* the Python nn.Module mirror in
 * maskcyclegan-vc_amd/mask_cyclegan_vc/: every entry point names the reference interface it
 * replaces (file:line relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers and sizes; every pointer is a DEVICE pointer to contiguous fp32 unless noted;
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); all calls are asynchronous w.r.t.
 *     the host and allocate no device memory on the compute path.  Process-wide state the library DOES keep
 *     (all behind mutexes): a pool of timing-less hipEvents for the lane <-> auxiliary-stream hand-offs, per-device
 *     weight-pack job tables (one small hipMalloc each on first use, outside stream capture), the cached workspace
 *     plans per (network, B, T), the deterministic-mode switch and the opt-in trace log.  Calls on different streams
 *     may be issued from different host threads, EXCEPT while the trace is enabled (its log is not thread-safe);
 *   - return value 0 = success, otherwise a hipError_t value or MCVC_ERR_* (>= 1000); the Python
 *     wrapper raises RuntimeError;
 *   - parameter tables are arrays of device pointers in the reference's `named_parameters()` order
 *     (Generator: 110 tensors, Discriminator: 20 tensors); gradient tables use
 *     the same order, entries may be NULL (skipped); gradients are ACCUMULATED (+=);
 *   - `packed` buffers hold the K-major re-packed weights the MFMA kernels read; they must be
 *     zero-initialised once by the caller and refreshed with mcvc_*_pack after every parameter update;
 *   - `stash` keeps what backward needs (conv outputs, InstanceNorm statistics, activations);
 *     `scratch` is transient (split-K slabs, gradient ping-pong buffers).
 */
#ifndef MCVC_H
#define MCVC_H

#ifdef __cplusplus
extern "C" {
#endif

#define MCVC_ABI_VERSION 3          /* r4: + mcvc_gen_backward_prefix, mcvc_set_trunk_passes_in_flight, op-level Winograd / staged-GEMM / trunk entries;
                                       3: + mcvc_gen_update_ranges / mcvc_disc_update_batch (optimizer step fused with the re-pack) */
#define MCVC_GEN_NPARAMS 110
#define MCVC_DISC_NPARAMS 20
#define MCVC_N_MEL 80

int mcvc_version(void);

/* Deterministic mode (default: env MCVC_DETERMINISTIC, else off).  When on, every reduction that would otherwise use
 * floating-point atomics (K-split accumulate of a conv / trunk data-gradient, the tiny-dW weight gradients) takes a
 * fixed-order path (private slabs summed by the consumer), so a step is bit-reproducible run to run like the
 * reference's CPU path.  Workspaces are always sized for both modes.  Returns the previous setting.             */
int mcvc_set_deterministic(int on);
int mcvc_get_deterministic(void);
/* Precise mode (default: env MCVC_PRECISE, else off; r6).  When on, the generators' 5x5 convolutions (model.py:86-99 downSample1/2,
 * :192-204 upSample1/2) run on the direct im2col-free kernels in every pass -- no Winograd scheme -- so the step's rounding is that of
 * plain fp32 sums, the reference's own (F.conv2d): against an fp64 run of the same four iterations the parameters then sit as close as
 * the reference's CPU fp32 arithmetic does (tests/test_hip_parity_fp64.py).  The default (fast) mode is 2-3x noisier on that measure and
 * inside 1e-3 on every single-iteration quantity.  Process-wide; set it before packing networks / building engines.  Returns the
 * previous setting.                                                                                                              */
int mcvc_set_precise(int on);
int mcvc_get_precise(void);
/* Small-batch generator passes (B * T/4 <= 32) run the 1-D trunk (six residual blocks + conv1dto2d, model.py:258-271) as ONE
 * persistent launch per direction instead of one fused launch per layer (default on).  Results are
 * bit-identical either way; the switch exists for A/B timing and tests.  Returns the previous setting.
 * The BACKWARD pass's data-gradient chain through the six blocks is one persistent launch as well (B * T/4 <= 16 at the 1024 staged
 * channels): on = 1 both directions, 2 forward only, 0 neither.  The persistent backward sums every K range in a
 * fixed order, so -- unlike the per-layer launches' atomic K-split -- its gradients are bit-reproducible in every mode; they agree with the
 * per-layer path to rounding.                                                                                  */
int mcvc_set_trunk_persistent(int on);

/* ---- grouped launches: two networks of the same architecture in one grid (new; the reference runs G_A2B / G_B2A and the discriminator
 *      pairs one after the other, train.py:203-216, 255-273).  Bracket TWO identical sequences of library calls that differ in pointers
 *      only (weights, activations, workspaces, destinations -- same B, T, flags, streams):
 *          mcvc_twin_begin();   <sequence on network 0>   mcvc_twin_switch();   <the same sequence on network 1>   rc = mcvc_twin_end();
 *      Between begin and switch nothing reaches the stream (the launches are recorded); the second sequence launches every kernel ONCE
 *      with gridDim.z = 2, one z-slice per network: half the launches, twice the workgroups per launch.  Any call that launches kernels
 *      may appear inside (passes, losses, packs, Adam); calls keep their meaning and return codes.  mcvc_twin_end returns MCVC_ERR_INVALID
 *      if the two sequences did not issue the same kernels with the same grids (nothing is repaired: treat the buffers as undefined).
 *      The bracket is per host thread; do not enqueue other work on the same streams between begin and switch (it would run BEFORE the
 *      grouped kernels).  The trace (mcvc_trace_*) counts a grouped launch once, with both networks' FLOPs and bytes.      */
int mcvc_twin_begin(void);
int mcvc_twin_switch(void);
int mcvc_twin_end(void);
int mcvc_twin_launches(void);           /* kernels recorded by the last bracket of this thread */

/* ---- sizes (floats) -------------------------------------------------------------------------- */
long long mcvc_gen_packed_floats(void);
long long mcvc_disc_packed_floats(void);
long long mcvc_gen_stash_floats(int B, int T);
long long mcvc_gen_scratch_floats(int B, int T);
long long mcvc_disc_stash_floats(int B, int T);
long long mcvc_disc_scratch_floats(int B, int T);
int mcvc_gen_out_frames(int T);          /* T' of Generator.forward (== T when T % 4 == 0) */
int mcvc_disc_out_frames(int T);         /* last dim of Discriminator.forward (T/8 when T % 8 == 0) */

/* ---- weight packing (after every optimizer step / load_state_dict) --------------------------- */
int mcvc_gen_pack(const float* const* params, float* packed, void* stream);
/*      Small-batch variant: when every pass until the next re-pack has batch <= max_batch at n_frames T and
 *      mcvc_gen_trunk_fused(max_batch, T) is 1, the 1-D trunk runs on the fused trunk kernels, which read the OIHW
 *      parameters directly -- the generic K-major copies of the trunk layers (60 % of the generator's weights) are then
 *      not refreshed.  Falls back to the full re-pack otherwise.                                                   */
int mcvc_gen_trunk_fused(int B, int T);
int mcvc_gen_pack_small_batch(const float* const* params, float* packed, int max_batch, int T, void* stream);
/* The same in two parts: sets = 1 refreshes only what a FORWARD pass reads (K-major forward copies, biases, forward Winograd sets),
 * sets = 2 only what a BACKWARD pass reads (data-gradient / transposed / data-gradient Winograd sets), 3 = both.  After sets = 1 a
 * backward pass on this buffer returns MCVC_ERR_INVALID until sets = 2 has run (the trainer's discriminator phase needs the updated
 * generators forward-only; the rest of the refresh runs beside it).                                    */
int mcvc_gen_pack_sets(const float* const* params, float* packed, int max_batch, int T, int sets, void* stream);
/* ... and restricted to parameter ranges (range_mask bit 0: parameters [100,110), bit 1: [24,100), bit 2: [0,24); 7 = all; r4: bits 3 / 4 / 5
 * = [12,24) / [4,12) / [0,4), the head in three parts -- see MCVC_BWD_FINE_MILESTONES): the ranges
 * whose gradients mcvc_gen_backward_overlap reports final one after the other, so that a range's optimizer step + re-pack can run beside the
 * rest of the backward pass.                                                                              */
int mcvc_gen_pack_ranges(const float* const* params, float* packed, int max_batch, int T, int sets, int range_mask, void* stream);
int mcvc_disc_pack(const float* const* params, float* packed, void* stream);
/*      The same restricted to what passes at n_frames T read when the three stride-2 layers run as implicit GEMMs (every batch size
 *      since r5): their tap-major K-major forward copy + biases only -- it serves the forward pass and, read row-major, the data gradient;
 *      the weight gradient reads no weights.  Falls back to the full re-pack when a layer would not take that path (planes whose width is
 *      no multiple of 8).  A pass that would need one of the stale copies returns MCVC_ERR_INVALID instead of reading it.                 */
int mcvc_disc_pack_small(const float* const* params, float* packed, int T, void* stream);
/*      ... for passes of up to max_batch samples (same copies at every max_batch since r5; the argument is kept for ABI 3).            */
int mcvc_disc_pack_batch(const float* const* params, float* packed, int max_batch, int T, void* stream);

/* ---- Generator: replaces Generator.forward (mask_cyclegan_vc/model.py:239-280) and its autograd
 *      x, mask: [B,80,T] (mask NULL = all ones, test.py:92); out: [B,80,T']                        */
int mcvc_gen_forward(const float* const* params, const float* packed, const float* x, const float* mask,
                     float* out, float* stash, float* scratch, long long scratch_floats, int B, int T, void* stream);
/*      dout: [B,80,T'] ; dx (nullable): [B,80,T], written or accumulated (accumulate_dx != 0)      */
int mcvc_gen_backward(const float* const* params, const float* packed, float* const* grads, const float* mask,
                      const float* dout, float* dx, int accumulate_dx, const float* stash,
                      float* scratch, long long scratch_floats, int B, int T, void* stream, void* aux_stream);
/*      aux_stream (nullable): a second hipStream_t on which the weight-gradient kernels run beside the data-gradient
 *      chain; the call returns with `stream` ordered after everything launched on aux_stream.            */
/*      Same, plus gradient-ready milestones for overlapping a data-parallel all-reduce with the rest of the pass
 *      (new: the reference has no distributed code).  milestones (nullable): two caller-owned hipEvent_t handles;
 *      [0] is recorded once every gradient of parameters [100,110) (upSample1/2 + lastConvLayer; named_parameters()
 *      order) has been produced, [1] once those of [24,100) (six residual blocks + conv1dto2d) have; the remaining
 *      [0,24) are complete when the call's work on `stream` is.  A consumer stream waits on the event.       */
int mcvc_gen_backward_overlap(const float* const* params, const float* packed, float* const* grads, const float* mask,
                              const float* dout, float* dx, int accumulate_dx, const float* stash,
                              float* scratch, long long scratch_floats, int B, int T, void* stream, void* aux_stream,
                              void* const* milestones);

/*      The small-batch passes run the residual trunk as ONE persistent launch whose 64 workgroups hand activations to each other inside
 *      the kernel (bounded spins).  Should a workgroup ever give up (the device cannot keep all 64 resident: more than four such passes in
 *      flight at once), the pass's output is poisoned with NaN -- so every loss of the step turns non-finite -- and an error word in the
 *      scratch buffer records the layer.  mcvc_gen_trunk_fault reads that word (0 = none, 1 + layer otherwise, negative = call failed) and
 *      optionally clears it; it SYNCHRONISES `stream`.  Call it with reset != 0 once after allocating a scratch buffer (the word is not
 *      initialised by the passes), then e.g. once per logging interval.  mcvc_debug_trunk_fault_inject(1) makes the following persistent
 *      launches lose one arrival (test hook for exactly this path); returns the previous setting.             */
int mcvc_gen_trunk_fault(float* scratch, int B, int T, int reset, void* stream);
int mcvc_debug_trunk_fault_inject(int on);

/*      Same with flags.  MCVC_BWD_NO_JOIN (1): return WITHOUT ordering `stream` after the weight-gradient kernels still running on aux_stream
 *      (the data-gradient result dx is complete on `stream`).  The caller then owes: (i) the next pass that accumulates into the same gradient
 *      tensors uses the SAME aux_stream (in-order) or waits for it, (ii) `scratch` and `stash` of this pass stay untouched until aux_stream
 *      has drained (the next pass uses other buffers), (iii) a later pass on that aux_stream joins (flags = 0) before the gradients are read.
 *      The trainer runs the cycle pass's backward this way: the translation pass's data-gradient chain starts ~0.2 ms earlier.       */
#define MCVC_BWD_NO_JOIN 1
/*      flags & MCVC_BWD_FINE_MILESTONES (r4): `milestones` holds FOUR events; [2] is recorded when the gradients of parameters [12,24)
 *      (downSample2, conv2dto1d) are final, [3] when those of [4,12) (downSample1) are: the head of the network in the order a backward pass
 *      finishes it, so that the optimizer step of all but conv1 (mcvc_gen_update_ranges, range_mask bits 8 / 16 / 32 = [12,24) / [4,12) /
 *      [0,4); bit 4 = their union, not to be combined with them) runs beside the rest of the pass instead of behind it.                */
#define MCVC_BWD_FINE_MILESTONES 2
int mcvc_gen_backward_flags(const float* const* params, const float* packed, float* const* grads, const float* mask,
                            const float* dout, float* dx, int accumulate_dx, const float* stash,
                            float* scratch, long long scratch_floats, int B, int T, void* stream, void* aux_stream,
                            void* const* milestones, int flags);
/*      Backward over a PREFIX of a forward pass's samples (r4).  `stash` was written by mcvc_gen_forward with batch stash_B >= B; the pass
 *      back-propagates through its first B samples only (mask / dout / dx hold B samples; scratch is sized for B).  The trainer batches
 *      the discriminator phase's generator forwards of iteration t (train.py:259-273: outputs only, their backward is discarded) into the
 *      generator phase's forwards of iteration t+1 (train.py:203-210: same weights), so those passes run over 3 (2) samples and their
 *      backward over the first 2 (1).  The 2-D stash tensors are batch-major; the 1-D trunk's are [C][stash_B][T/4] and are read with that
 *      channel pitch.  stash_B == B is mcvc_gen_backward_flags.                                                                      */
int mcvc_gen_backward_prefix(const float* const* params, const float* packed, float* const* grads, const float* mask,
                             const float* dout, float* dx, int accumulate_dx, const float* stash, int stash_B,
                             float* scratch, long long scratch_floats, int B, int T, void* stream, void* aux_stream,
                             void* const* milestones, int flags);
/*      ... over the WINDOW [stash_b0, stash_b0 + B) of the forward pass's samples (stash_b0 * T/4 must be a multiple of 4).  The trainer
 *      back-propagates the identity sample of a merged pass (train.py:223-224: depends on nothing but its own forward) while the
 *      discriminators of the cycle chain are still running, and the translation sample -- the window [1, 2) -- at the end of the chain.  */
int mcvc_gen_backward_window(const float* const* params, const float* packed, float* const* grads, const float* mask,
                             const float* dout, float* dx, int accumulate_dx, const float* stash, int stash_B, int stash_b0,
                             float* scratch, long long scratch_floats, int B, int T, void* stream, void* aux_stream,
                             void* const* milestones, int flags);
/*      Number of persistent trunk passes the caller keeps in flight at once (a grouped launch counts as two; default 2).  The persistent
 *      kernels' 64 workgroups per pass wait for each other inside the kernel, so all of them must be resident: they are used only while
 *      64 x that number <= the device's compute units, otherwise the passes fall back to per-layer launches.  Returns the previous value. */
int mcvc_set_trunk_passes_in_flight(int n);
/*      bit 0 / bit 1: a (B, T) generator pass would run its forward / backward trunk on the persistent kernels under the current
 *      switches and residency bound (0: per-layer launches).                                                                          */
int mcvc_gen_trunk_persistent(int B, int T);

/* ---- Generator inference in bf16 (BASELINE configs[4]: generator_A2B, bs=16, 80 x 512 frames).  Replaces the call
 *      `generator(real, ones_like(real))` of the reference's inference driver (mask_cyclegan_vc/test.py:92, 107 ->
 *      Generator.forward, model.py:239-280) when the caller asks for bf16: NHWC bf16 activations, bf16 MFMA with fp32
 *      accumulation, fp32 InstanceNorm statistics; x / mask / out stay fp32 [B,80,T] at the boundary.
 *      packed: mcvc_gen_bf16_packed_bytes() bytes, refreshed with mcvc_gen_bf16_pack after every parameter change
 *      (weights are cast from the fp32 parameters); workspace: mcvc_gen_bf16_workspace_bytes(B, T) bytes, 256-byte aligned;
 *      mask NULL = all ones.  No gradient path.                                                               */
long long mcvc_gen_bf16_packed_bytes(void);
long long mcvc_gen_bf16_workspace_bytes(int B, int T);
int mcvc_gen_bf16_pack(const float* const* params, void* packed, void* stream);
int mcvc_gen_infer_bf16(const float* const* params, const void* packed, const float* x, const float* mask, float* out, void* workspace,
                        long long workspace_bytes, int B, int T, void* stream);

/*      single-op entry points of the bf16 path (kernel parity tests; same kernels the forward uses).  Tensors are NHWC bf16
 *      (raw 16-bit storage = torch.bfloat16): y[N][OH][OW][Cout] = conv2d(x[N][H][W][Cin], w[Cout][Cin][KH][KW] fp32 -> bf16) + bias;
 *      Cin % 32 == 0, Cout % 4 == 0; wpack: mcvc_bf16_conv2d_pack_bytes() bytes of scratch.                      */
long long mcvc_bf16_conv2d_pack_bytes(int Cout, int Cin, int KH, int KW);
int mcvc_bf16_conv2d(const void* x, const float* w, const float* bias, void* y, void* wpack, int N, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad_h, int pad_w, void* stream);
/*      conv1 of the bf16 forward with its input preparation and gated GLU fused (r6; model.py:241-242):
 *      y[B][80][T][128] bf16 NHWC = conv2d(stack(x * mask, mask), w, b, padding (2, 7)) * sigmoid(conv2d(..., wg, bg)); x, mask fp32 [B][80][T]
 *      (mask NULL = ones), w / wg [128][2][5][15], b / bg [128] fp32; wpack: mcvc_bf16_conv1_glu_pack_bytes() bytes, 16-byte aligned.     */
long long mcvc_bf16_conv1_glu_pack_bytes(void);
int mcvc_bf16_conv1_glu(const float* x, const float* mask, const float* w, const float* b, const float* wg, const float* bg, void* y, void* wpack,
                        int B, int T, void* stream);
/*      the generator's last conv (r6; model.py:207-211, 278-279) through the fused kernel of the bf16 forward: out[B][80][T] fp32 =
 *      conv2d(x[B][80][T][128] bf16 NHWC, w[1][128][5][15], b[1], padding (2, 7)); wpack: mcvc_bf16_last_conv_pack_bytes() bytes, 16-byte aligned. */
long long mcvc_bf16_last_conv_pack_bytes(void);
int mcvc_bf16_last_conv(const void* x, const float* w, const float* b, float* out, void* wpack, int B, int T, void* stream);
/*      conv2dto1d (Conv1d 5120 -> 256, k = 1) + conv2dto1dLayer_tfan (r6; model.py:142-146, 254-255) through the fused kernel of the bf16 forward:
 *      x [B][W][5120] bf16 whose memory channel h * 256 + c holds the reference's channel c * 20 + h (the view(B, 5120, 1, -1) of :249-251 as the
 *      forward lays it out), w [256][5120] fp32 in the reference's order, gamma / beta [256]; y [B][W][256] bf16.  W <= 128 else MCVC_ERR_INVALID.  */
long long mcvc_bf16_c2d1d_pack_bytes(void);
int mcvc_bf16_c2d1d_norm(const void* x, const float* w, const float* gamma, const float* beta, void* y, void* wpack, int B, int W, void* stream);
/*      one layer of a residual block (r6; model.py:47-76) through the fused kernel of the bf16 forward -- Conv1d(k = 3, padding 1) + InstanceNorm1d
 *      (affine) + {value * sigmoid(gate) when w_gate != NULL | + residual}: x [B][W][Cin], y / residual [B][W][C] bf16 (channel-innermost);
 *      w / w_gate [C][Cin][3], gamma / beta (+ the gate's) [C] fp32.  W <= 128, Cin = 256 or 512, C % 32 == 0, else MCVC_ERR_INVALID
 *      (the forward then runs conv + norm as two launches).  wpack: mcvc_bf16_trunk_layer_pack_bytes() bytes, 16-byte aligned.               */
long long mcvc_bf16_trunk_layer_pack_bytes(int Cin, int C, int gated);
int mcvc_bf16_trunk_layer(const void* x, const float* w, const float* w_gate, const float* gamma, const float* beta, const float* gamma_gate,
                          const float* beta_gate, const void* residual, void* y, void* wpack, int B, int W, int Cin, int C, void* stream);
/*      y = act(InstanceNorm(x)) (+ residual): act 0 none, 1 gated GLU (Cx = 2C: value | gate), 2 x*sigmoid(x); pixel_shuffle != 0:
 *      the normalised tensor is PixelShuffle(2)(x), output [N][2H][2W][Cx/4].  scratch: N * 65 * Cx * 2 floats.      */
int mcvc_bf16_instnorm_act(const void* x, const float* gamma, const float* beta, const float* gamma_gate, const float* beta_gate,
                           const void* residual, void* y, float* scratch, int N, int H, int W, int Cx, int act, int pixel_shuffle, void* stream);

/* ---- Discriminator: replaces Discriminator.forward (model.py:340-349) and its autograd
 *      x: [B,80,T]; out: [B,1,10,T8] sigmoid probabilities                                          */
int mcvc_disc_forward(const float* const* params, const float* packed, const float* x, float* out,
                      float* stash, float* scratch, long long scratch_floats, int B, int T, void* stream);
/*      dout: [B,1,10,T8]; grad wrt the sigmoid OUTPUT (dout_is_logit_grad == 0) or wrt the pre-sigmoid
 *      logits (!= 0, what mcvc_lsgan_loss emits).  grads NULL = data-gradient only (generator phase,
 *      train.py:211-216 -- the reference computes D weight grads there and discards them).          */
int mcvc_disc_backward(const float* const* params, const float* packed, float* const* grads,
                       const float* dout, int dout_is_logit_grad, float* dx, int accumulate_dx,
                       const float* stash, float* scratch, long long scratch_floats, int B, int T, void* stream,
                       void* aux_stream);

/* ---- losses (train.py:219-237, 276-294). loss_slot/term_slot: device floats, accumulated (+=) -- */
/* loss_slot += weight*mean|a-b|, term_slot += mean|a-b|, grad_a (=|+=) weight*sign(a-b)/n           */
int mcvc_l1_loss(const float* a, const float* b, long long n, float weight, float* loss_slot, float* term_slot,
                 float* grad_a, int accumulate_grad, void* stream);
/* d = discriminator sigmoid output; loss_slot += weight*mean((target-d)^2);
 * grad_logit = d(loss)/d(pre-sigmoid logit)                                                         */
int mcvc_lsgan_loss(const float* d, long long n, float target, float weight, float* loss_slot, float* term_slot,
                    float* grad_logit, void* stream);

/* The terms of one phase may be produced on different streams, each into its own pair pairs[2k] = weight*mean, pairs[2k+1] = mean
 * (pass them as loss_slot / term_slot above, zeroed before).  This adds pair k to slots[loss_dst[k]] / slots[term_dst[k]] (-1 =
 * skip), k = 0..n-1 in that order (n <= 16; the two index arrays are HOST memory): the order of the reference's sums
 * (train.py:233-237, 276-294) whatever the streams' timing was.                                       */
int mcvc_loss_combine(const float* pairs, int n, const int* loss_dst, const int* term_dst, float* slots, void* stream);

/* ---- optimizer: torch.optim.Adam(betas, eps, weight_decay=0) on a flat buffer (train.py:119-122) */
int mcvc_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/*      The same step on the gradient g + g2 (g2 nullable: a second buffer that independent backward passes accumulated into, so that they need
 *      not be ordered against the passes that write g), optionally clearing the gradient buffer(s) behind the read (zero_grads != 0: what
 *      reset_grad / zero_grad, train.py:157-161, does before the next accumulation -- here without a separate pass over the buffers).   */
int mcvc_adam_step2(float* p, float* g, float* g2, int zero_grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                    float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/*      optimizer.step() fused with the weight re-pack (reference train.py:242 / :299 are just `optimizer.step()`: the packed copies are this
 *      library's own, so their refresh belongs to the step): ONE launch in which a workgroup owns a tile of filters of one parameter tensor --
 *      it applies the Adam update above to the tile (bit-identical to mcvc_adam_step2), keeps the new weights in LDS and writes every packed
 *      copy derived from them (K-major / tap-major / data-gradient / transposed trunk / Winograd U sets) from there.  No second
 *      pass over the OIHW tensors: the per-step `pack` kernel family is gone (it remains for load_state_dict: mcvc_*_pack*).
 *      numel[i]: elements of parameter tensor i (0: a parameter that is neither updated nor packed -- the unused downSample4 block);
 *      flat / grad / grad2 (nullable) / exp_avg / exp_avg_sq: flat buffers in which params[i], its gradient(s) and moments sit at EQUAL
 *      offsets (gradient of params[i] = grad + (params[i] - flat)); range_mask / max_batch / T as mcvc_gen_pack_ranges / mcvc_disc_pack_batch. */
int mcvc_gen_update_ranges(const float* const* params, const long long* numel, float* packed, int max_batch, int T, int range_mask,
                           const float* flat, float* grad, float* grad2, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                           float beta2, float eps, int step, float grad_scale, int zero_grads, void* stream);
int mcvc_disc_update_batch(const float* const* params, const long long* numel, float* packed, int max_batch, int T,
                           const float* flat, float* grad, float* grad2, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                           float beta2, float eps, int step, float grad_scale, int zero_grads, void* stream);
int mcvc_axpy(float* y, const float* x, float alpha, long long n, void* stream);

/* ---- on-device input pipeline: replaces VCDataset.__getitem__ + DataLoader collate + 4 H2D copies per iteration
 *      (dataset/vc_dataset.py:19-77, mask_cyclegan_vc/train.py:82-96, 187-190) with one launch per minibatch.
 *      bank_X: [80][frames_X] fp32, all utterances of speaker X side by side along the frame axis; offs_X: int32 [n_X+1]
 *      first frame of each utterance (every utterance >= T frames).  Per sample and speaker: utterance ~ U{0..n-1},
 *      crop lo ~ U{0..len-T}, mask size ~ U{0..max_mask_len-1}, start ~ U{0..T-size-1} -- the reference's distributions
 *      (vc_dataset.py:33-70) from a counter-based SplitMix64 stream keyed by (seed, step, sample, speaker): restated
 *      bit-exactly in oracle/sampler_oracle.py.  Outputs [B][80][T]; draws (nullable): int32 [B][2][4] =
 *      (utterance, lo, size, start).                                                                        */
int mcvc_draw_batch(const float* bank_A, const int* offs_A, int n_A, long long frames_A, const float* bank_B, const int* offs_B, int n_B,
                    long long frames_B, int B, int T, int max_mask_len, unsigned long long seed, unsigned long long step,
                    float* real_A, float* mask_A, float* real_B, float* mask_B, int* draws, void* stream);

/* ---- single-op entry points (kernel parity tests; same kernels the network calls use) ---------- */
/* y[N,Cout,OH,OW] = conv2d(x[N,Cin,H,W], w[Cout,Cin,KH,KW]) + bias ; stride 1 or 2.
 * wpack: scratch of mcvc_conv2d_pack_floats() floats, zero-initialised by the caller.
 * slabs: optional split-K scratch (max_slabs-1)*N*Cout*OH*OW floats; the reduced result is in y.     */
long long mcvc_conv2d_pack_floats(int Cout, int Cin, int KH, int KW);
int mcvc_conv2d_forward(const float* x, const float* w, const float* bias, float* y, float* wpack,
                        float* slabs, int max_slabs, int N, int Cin, int H, int W, int Cout, int KH, int KW,
                        int stride, int pad_h, int pad_w, int pixel_shuffle, void* stream);
/* dx[N,Cin,H,W] = conv2d_backward_data(dy[N,Cout,OH,OW], w) */
int mcvc_conv2d_dgrad(const float* dy, const float* w, float* dx, float* wpack, float* slabs, int max_slabs,
                      int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                      void* stream);
/* dw[Cout,Cin,KH,KW] += conv2d_backward_weight(x, dy).  slabs: K-split scratch of
 * mcvc_conv2d_wgrad_slab_floats() floats (may be NULL: no split)                                      */
long long mcvc_conv2d_wgrad_slab_floats(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_h, int pad_w);
int mcvc_conv2d_wgrad(const float* x, const float* dy, float* dw, float* slabs, long long slab_floats, int N, int Cin, int H, int W,
                      int Cout, int KH, int KW, int stride, int pad_h, int pad_w, void* stream);
/* ---- one convolution LAYER of the path, op by op (r4).  SURVEY.md section 8(b) asks for single-op entries of the kernels that are hot at
 *      the trainer's shapes: the 5x5 layers run as Winograd products (model.py:86-103 downSample: F(2x2,3x3) / F(4x4,3x3) over the four input
 *      phases; :226-237 upSample: F(2x2,5x5) / F(4x4,5x5)) and the discriminators' stride-2 3x3 layers (:298-314) as implicit GEMMs.  These
 *      entries run the planner the networks use (conv_fwd / conv_dgrad / conv_wgrad) on ONE layer of `branches` (1, or 2 = value | gate)
 *      convolutions Cin -> Cout:  x [N][Cin][H][W], y / dy [N][branches*Cout][OH][OW] (value rows first), dx like x, dw like the OIHW
 *      parameters (ACCUMULATED: pass zeros).  `packed` = mcvc_layer_pack of the parameters (every weight set the planner may read);
 *      `scratch` >= mcvc_layer_scratch_floats.  `scheme`: 0 the planner's choice at this shape, 1 Winograd with 2x2 output tiles, 2 Winograd
 *      with 4x4 output tiles (sample / tile thresholds lifted; image sides must be multiples of 4 -- 8 for the stride-2 layers),
 *      3 no Winograd (direct kernels for these layers since r5), 4 direct kernels only, 5 implicit GEMM (forward / dgrad / wgrad of the 3x3
 *      stride-2 layers: the kernels the discriminator passes run, sgemm.h; the dense operand is converted to their layout first).  w0 / w1 = the OIHW tensors themselves (the
 *      pre-r5 staged data gradient multiplied them in place; unused now).  pixel_shuffle: the forward store of the up-sampling layers (model.py:232).   */
long long mcvc_layer_packed_floats(int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w);
long long mcvc_layer_scratch_floats(int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w);
int mcvc_layer_pack(const float* w0, const float* b0, const float* w1, const float* b1, float* packed, int Cin, int Cout, int branches,
                    int KH, int KW, int stride, int pad_h, int pad_w, void* stream);
int mcvc_layer_forward(const float* x, const float* packed, const float* w0, const float* w1, float* y, float* scratch,
                       long long scratch_floats, int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride,
                       int pad_h, int pad_w, int scheme, int pixel_shuffle, void* stream);
int mcvc_layer_dgrad(const float* dy, const float* packed, const float* w0, const float* w1, float* dx, float* scratch,
                     long long scratch_floats, int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride,
                     int pad_h, int pad_w, int scheme, void* stream);
int mcvc_layer_wgrad(const float* x, const float* dy, float* dw0, float* dw1, float* scratch, long long scratch_floats, int N, int H,
                     int W, int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w, int scheme, void* stream);
/*      The fused backward of one 1-D trunk layer (SURVEY.md section 8b resblock1d_bwd / gemm1x1_in_bwd; reference model.py:47-76 under
 *      autograd): InstanceNorm (+ gated GLU when the gate pointers are given) backward of dy [Cout][B][T4] recomputed inside the
 *      transposed-convolution launch; dconv [Cx][B][T4] = gradient w.r.t. the conv output (Cx = Cout or 2*Cout, value rows first);
 *      dx [Cin][B][T4] += data gradient; d(gamma), d(beta) += (nullable); x_in != NULL (k = 3): dw / dw_gate += weight gradients.
 *      wpack: Cin * Cx * KW floats of workspace (the transposed weight copy).                                                          */
int mcvc_trunk_layer_backward(const float* dy, const float* conv_out, const float* stats, const float* gamma, const float* beta,
                              const float* gamma_gate, const float* beta_gate, const float* w, const float* w_gate, const float* x_in,
                              float* dx, float* dconv, float* dgamma, float* dbeta, float* dgamma_gate, float* dbeta_gate, float* dw,
                              float* dw_gate, float* wpack, int B, int Cin, int T4, int Cout, int KW, void* stream);

/* InstanceNorm(affine) + activation.  act: 0 none, 1 gated GLU (x has 2C channels: value|gate), 2 SiLU.
 * x[N,Cx,H,W] -> y[N,C,H,W]; stats[N,Cx,2] (mean, rstd)                                              */
int mcvc_instnorm_act_forward(float* x, const float* gamma, const float* beta, const float* gamma_gate,
                              const float* beta_gate, const float* residual, float* y, float* stats,
                              int N, int C, int H, int W, int act, void* stream);
int mcvc_instnorm_act_backward(const float* x, const float* gamma, const float* beta, const float* gamma_gate,
                               const float* beta_gate, const float* stats, float* dy, float* dx,
                               float* dgamma, float* dbeta, float* dgamma_gate, float* dbeta_gate,
                               int N, int C, int H, int W, int act, void* stream);

/* Fused 1-D trunk layer at small batch (B * T4 <= 32): ONE launch for Conv1d(k = 1 | 3, padding (k-1)/2) + bias + InstanceNorm1d(affine)
 * + {gated GLU when w_gate != NULL | residual add | nothing} -- the building block of ResidualLayer.forward (model.py:47-76:
 * value|gate pair, then the output conv with the skip connection) and of conv1dto2dLayer + its norm (model.py:266-267) -- SURVEY.md
 * section 8b's resblock1d / gemm1x1_in ops.  Trunk layout, channel-major with the batch inside: x [Cin][B][T4], w [Cout][Cin][KW] (the
 * nn.Conv1d tensor as is), conv_out [Cout (x2 with a gate)][B][T4] (pre-norm, kept for backward), stats [B][Cout (x2)][2] = (mean, rstd),
 * y / residual [Cout][B][T4].                                                                                         */
int mcvc_trunk_layer_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, const float* w_gate,
                             const float* bias_gate, const float* gamma_gate, const float* beta_gate, const float* residual, float* conv_out,
                             float* stats, float* y, int B, int Cin, int T4, int Cout, int KW, void* stream);

/* Batched fp32 GEMM of the Winograd convolution paths (the 36 / 16 per-point products of upSample1/2 and downSample1/2, forward,
 * data-gradient and weight-gradient: the kernel family with the largest share of a bs=1 step).  For x in [0, nbatch):
 *   C_x[m][n] = sum_k A_x[k][m] * B_x[k][n]     A_x = a + x*a_stride, K-major [K][lda]; B_x [K][ldb]; C_x [M][ldc].
 * M % 128 == 0, K % 16 == 0, lda / ldb % 4 == 0, ldb >= 64, N <= ldb.  Exact fp32 (v_mfma_f32_32x32x2_f32).            */
int mcvc_batched_gemm(const float* a, const float* b, float* c, int nbatch, int M, int N, int K, int lda, int ldb, int ldc,
                      long long a_stride, long long b_stride, long long c_stride, void* stream);

/* db[C] += sum over (n, h, w) of dy[N,C,P] */
int mcvc_bias_grad(const float* dy, float* db, int N, int C, int P, void* stream);
/* norm-less activations: act 1 gated GLU (x[N,2C,P] -> y[N,C,P]), 2 x*sigmoid(x), 3 sigmoid */
int mcvc_act_forward(float* x, float* y, int N, int C, int P, int act, void* stream);
int mcvc_act_backward(const float* x, float* dy, float* dx, int N, int C, int P, int act, void* stream);
/* xin[N,2,P] = (x*mask, mask)   (model.py:241) ; dx (=|+=) dxin[:,0]*mask */
int mcvc_fif_input(const float* x, const float* mask, float* xin, int N, int P, void* stream);
int mcvc_fif_input_grad(const float* dxin, const float* mask, float* dx, int N, int P, int accumulate, void* stream);

/* ---- opt-in measurement (bench.py `roofline`): HIP events around every kernel launch ----------- */
int mcvc_trace_enable(int on);
int mcvc_trace_kinds(void);
const char* mcvc_trace_kind_name(int kind);
/* out[kind][4] = {launches, total_ms, algorithmic_flops, algorithmic_bytes}; clears the log; syncs  */
int mcvc_trace_collect(double* out);
/* raw records in launch order: out[i][4] = {kind, ms, flops, bytes}; returns the record count           */
int mcvc_trace_collect_raw(double* out, int max_records);

#ifdef __cplusplus
}
#endif
#endif /* MCVC_H */
