#!/usr/bin/env python
"""Hardware counters of one tools/kbench.py target in as many SEPARATE `rocprofv3 --pmc` passes as there are groups (--kernel-trace only, the method of
tools/pmc_kbench.py), means over the last launches of every (kernel, grid) group whose name contains --match.

    python tools/pmc_groups.py wino63_boundary --match boundary_kernel --set cu --out gpurun_out/pmc_boundary.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_kbench as P   # noqa: E402

SETS = {
    "cu": [
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS",
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS",
        "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES",
        "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL",
        "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum",
        "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum",
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum",
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum",
        "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum",
        "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_LFIFO_STALL_CYCLES_sum",
        "TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum",
        "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TCP_GATE_EN2_sum",
        "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WINDOW_VALID SPI_RA_REQ_NO_ALLOC_CSN",
        "GRBM_GUI_ACTIVE FETCH_SIZE",
        "WRITE_SIZE",
    ],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target")
    ap.add_argument("--match", required=True)
    ap.add_argument("--set", default="cu")
    ap.add_argument("--opts", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--last", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    res = {}
    for ctr in SETS[a.set]:
        try:
            kt, cc = P.one_pass(ctr, a.target, a.opts, ["--iters", str(a.iters)], float(os.environ.get("PMC_TIMEOUT_S", "240")))
        except Exception as ex:
            print("pass failed: %s (%s)" % (ctr, ex), file=sys.stderr)
            continue
        for key, (ns, vals, n, _) in P.group(kt, cc, a.match, a.last).items():
            e = res.setdefault("%s | %s" % (key[0][:90], key[1]), {"ns": []})
            e["ns"].append(ns)
            e.update(vals)
    for k, e in res.items():
        e["avg_ns"] = sum(e["ns"]) / len(e["ns"])
        del e["ns"]
    txt = json.dumps(res, indent=1, sort_keys=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
