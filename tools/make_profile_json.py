#!/usr/bin/env python3
"""profiles/kernel_counters.json + profiles/hbm_traffic.json from a tools/prof_round.sh directory, tied to the kernels'
source hash (bench.py reports them only while that hash matches the sources it runs).
HBM bytes: the L2's memory-side request counters by size class (32 B x RDREQ_32B + 128 B x RDREQ_128B + 64 B x rest;
writes 64 B x WRREQ_64B + 32 B x rest) -- not FETCH_SIZE, which on gfx950 tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section).
    python tools/make_profile_json.py gpurun_out/prof_<tag> <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import _native  # noqa: E402


def main(d, tag, suffix=""):
    pmc = json.load(open(os.path.join(d, "pmc_summary.json")))
    bench = json.loads(open(os.path.join(d, "bench_under_stats.json")).read() or "{}")
    workload = bench.get("config", {}).get("workload", "")
    g = lambda k, n: pmc.get(k, {}).get(n, {}).get("mean_per_dispatch")  # noqa: E731
    kernels, traffic = {}, {}
    for k in pmc:
        e = {"valu_wave_insts_per_launch": g(k, "SQ_INSTS_VALU"), "lds_insts_per_launch": g(k, "SQ_INSTS_LDS"),
             "salu_insts_per_launch": g(k, "SQ_INSTS_SALU"), "waves": g(k, "SQ_WAVES"), "wave_quad_cycles": g(k, "SQ_WAVE_CYCLES"),
             "lds_idx_active_cycles": g(k, "SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict_cycles": g(k, "SQ_LDS_BANK_CONFLICT"),
             "wait_any": g(k, "SQ_WAIT_ANY"), "wait_inst_any": g(k, "SQ_WAIT_INST_ANY")}
        kernels[k] = e
        rd, rd32, rd128 = g(k, "TCC_EA0_RDREQ_sum"), g(k, "TCC_EA0_RDREQ_32B_sum"), g(k, "TCC_EA0_RDREQ_128B_sum")
        wr, wr64 = g(k, "TCC_EA0_WRREQ_sum"), g(k, "TCC_EA0_WRREQ_64B_sum")
        t = {}
        if None not in (rd, rd32, rd128):
            t["read_bytes"] = 32 * rd32 + 128 * rd128 + 64 * (rd - rd32 - rd128)
        if None not in (wr, wr64):
            t["write_bytes"] = 64 * wr64 + 32 * (wr - wr64)
        if len(t) == 2:
            t["hbm_bytes"] = t["read_bytes"] + t["write_bytes"]
        traffic[k] = t
    final = [k for k in pmc if "stage_tube" in k or "stage_up" in k or "tile_stat" in k or "site_flag" in k or "stage_slab" in k or "detail_" in k]
    tube = [k for k in pmc if "stage_tube" in k]
    first = [k for k in pmc if "stage_u1" in k]
    h = _native.source_hash()
    kc = {"source_hash": h, "workload": workload, "tag": tag, "frames": bench.get("config", {}).get("frames_per_launch", bench.get("config", {}).get("frames_per_gpu")), "kernels": kernels}
    if tube:
        kc["final_stage_kernel"] = dict(kernels[tube[0]], name=tube[0],
                                        lds_bytes_per_launch=None if kernels[tube[0]]["lds_insts_per_launch"] is None else
                                        kernels[tube[0]]["lds_insts_per_launch"] * 64 * 16 * 10.0 / 13.4)   # ~10 of 13.4 LDS instructions per pass are 16-byte row reads
    json.dump(kc, open(os.path.join(ROOT, "profiles", "kernel_counters%s.json" % suffix), "w"), indent=1)
    tr = {"source_hash": h, "workload": workload, "tag": tag, "kernels": traffic,
          "final_stage_bytes_per_launch": sum(traffic[k].get("hbm_bytes", 0) for k in final),
          "pipeline_bytes_per_step": sum(traffic[k].get("hbm_bytes", 0) for k in final + first)}
    json.dump(tr, open(os.path.join(ROOT, "profiles", "hbm_traffic%s.json" % suffix), "w"), indent=1)
    print(json.dumps({"source_hash": h, "final_stage_bytes_per_launch": tr["final_stage_bytes_per_launch"],
                      "pipeline_bytes_per_step": tr["pipeline_bytes_per_step"],
                      "tube_valu_insts": kc.get("final_stage_kernel", {}).get("valu_wave_insts_per_launch")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "round", sys.argv[3] if len(sys.argv) > 3 else "")
