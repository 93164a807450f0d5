#!/usr/bin/env python3
"""Per-kernel SQ summary from two rocprofv3 PMC passes (tools/profile_round.sh):  python tools/sq_summary.py passA.db passB.db

Derived per kernel (sums over all its dispatches; units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs, SQ_BUSY_CYCLES cycles summed over shader engines):
  mfma_busy   = MFMA_BUSY_CYCLES / (BUSY_CYCLES per SE x SIMDs)  -- share of the kernel's SIMD-cycles with the matrix pipe busy
  wait_any    = WAIT_ANY / WAVE_CYCLES      -- resident waves parked in s_waitcnt / s_barrier
  wait_inst   = WAIT_INST_ANY / WAVE_CYCLES -- issue stalls (MFMA read-after-write, pipe busy)
  active      = ACTIVE_INST_ANY / WAVE_CYCLES
  waves/disp  = SQ_WAVES per dispatch;  mfma/wave = INSTS_MFMA / WAVES
  lds_conf    = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE  -- extra LDS cycles from bank conflicts
"""
import re
import sqlite3
import sys


def load(path):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for k, c, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(k, {})[c] = (v, n)
    return out


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:58]


a, b = load(sys.argv[1]), load(sys.argv[2])
N_SE, N_SIMD = 32, 1024          # MI355X: 32 shader engines (8 XCD x 4), 256 CUs x 4 SIMDs
rows = []
for k, d in a.items():
    g = lambda c, dd=d: dd.get(c, (0.0, 1))[0]      # noqa: E731
    wc = g("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    nd = d["SQ_WAVE_CYCLES"][1]
    busy = g("SQ_BUSY_CYCLES") / N_SE                 # kernel-resident cycles (summed over dispatches)
    e = b.get(k, {})
    h = lambda c: e.get(c, (0.0, 1))[0]               # noqa: E731
    rows.append((busy, short(k), nd, g("SQ_WAVES") / nd, g("SQ_VALU_MFMA_BUSY_CYCLES") / max(busy * N_SIMD, 1.0), g("SQ_WAIT_ANY") / wc,
                 g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_INSTS_MFMA") / max(g("SQ_WAVES"), 1.0),
                 h("SQ_LDS_BANK_CONFLICT") / max(h("SQ_LDS_IDX_ACTIVE"), 1.0), h("SQ_INSTS_LDS") / max(h("SQ_INSTS_VALU"), 1.0)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("%-58s %6s %9s %9s %9s %9s %9s %9s %9s %8s %8s" % ("kernel (by share of busy cycles)", "disp", "share", "waves/d", "mfma_busy", "wait_any", "wait_inst", "active", "mfma/wave", "lds_conf", "lds/valu"))
for r in rows[:40]:
    print("%-58s %6d %8.1f%% %9.0f %9.3f %9.3f %9.3f %9.3f %9.1f %8.3f %8.3f" % (r[1], r[2], 100 * r[0] / tot, r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]))
