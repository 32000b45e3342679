#!/usr/bin/env python3
"""MFMA utilisation per kernel from the raw counter sums of profiles/collect_mfma.sh.

usage: mfma_util.py COUNTERS.json
For every kernel that issued fp64 MFMAs: MFMA busy cycles / GRBM_GUI_ACTIVE-normalised busy cycles, LDS bank-conflict
share, MFMA ops.  Counter semantics (MI355X_MICROARCH.md, constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
over the SIMDs that were sampled; SQ_BUSY_CYCLES the cycles an SQ had work; both are sums over the shader engines' SQs, so
the RATIO of two SQ counters of the same kernel is meaningful, absolute values are not compared across counters of
different blocks.  SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-flop units ("MOPS")."""
import json
import sys

d = json.load(open(sys.argv[1]))
rows = []
for k, c in d.items():
    g = lambda n: c.get(n, {}).get("sum_kb", 0.0)
    mf = g("SQ_VALU_MFMA_BUSY_CYCLES")
    if mf <= 0:
        continue
    rows.append((k, c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("calls", 0), mf, g("SQ_BUSY_CYCLES"), g("SQ_INSTS_VALU_MFMA_MOPS_F64"),
                 g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"), g("SQ_INSTS_VALU"), g("SQ_WAVE_CYCLES"), g("GRBM_GUI_ACTIVE")))
rows.sort(key=lambda r: -r[2])
# MI355X: 256 CUs x 4 SIMDs = 1024 SIMDs (SQ_VALU_MFMA_BUSY_CYCLES is summed over them), 8 XCDs (GRBM_GUI_ACTIVE is summed over
# them): utilisation = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8) = the share of the kernel's wall cycles in which a SIMD's matrix
# pipe was busy, averaged over the chip.  flops = MOPS_F64 x 512; TFLOP/s at the counters' clock = flops / (GUI_ACTIVE / 8) x f.
NSIMD, NXCD = 1024.0, 8.0
print("%-12s %6s %14s %14s %10s %14s %12s %14s %9s" % ("kernel", "calls", "MFMA_BUSY", "GUI_ACTIVE", "mfma_util", "MOPS_F64", "flop/cycle",
                                                     "LDS_CONFLICT", "confl/lds"))
for k, n, mf, sq, mops, lc, la, iv, wc, ga in rows:
    wall = ga / NXCD
    print("%-12s %6d %14.4g %14.4g %10.3f %14.4g %12.1f %14.4g %9.3f" % (k, n, mf, ga, (mf / NSIMD) / wall if wall else float("nan"), mops,
                                                                      mops * 512.0 / wall if wall else float("nan"), lc,
                                                                      lc / la if la else float("nan")))
print("(fp64 MFMA peak: 128 flop/cycle/CU x 256 CUs = 32768 flop/cycle = 78.6 TFLOP/s at 2.4 GHz; flop/cycle / 32768 = mfma_util, both clock-free)")
