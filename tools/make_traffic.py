#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a tools/prof_pmc.sh summary: HBM-side bytes per launch of the final-stage
kernel.  Read bytes come from the L2's memory-side request counters by size class
(32*RDREQ_32B + 128*RDREQ_128B + 64*rest) rather than FETCH_SIZE, which on gfx950 tallies 128-B requests
at 64 B (MI355X_MICROARCH.md, HBM section); write bytes = 64*WRREQ_64B + 32*rest (WRITE_SIZE agrees).

    python tools/make_traffic.py gpurun_out/pmc_<tag>/summary.json "<workload string printed by bench.py>"
"""
import json
import sys


def main(path, workload, out):
    d = json.load(open(path))
    rec = {"workload": workload, "source": path, "kernels": {}}
    for k, c in d.items():
        g = lambda n: c.get(n, {}).get("mean_per_dispatch")  # noqa: E731
        rd, rd32, rd128 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum"), g("TCC_EA0_RDREQ_128B_sum")
        wr, wr64 = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum")
        e = {"FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
             "TCC_HIT": g("TCC_HIT_sum"), "TCC_MISS": g("TCC_MISS_sum")}
        if None not in (rd, rd32, rd128):
            e["read_bytes"] = 32 * rd32 + 128 * rd128 + 64 * (rd - rd32 - rd128)
        if None not in (wr, wr64):
            e["write_bytes"] = 64 * wr64 + 32 * (wr - wr64)
        if "read_bytes" in e and "write_bytes" in e:
            e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
        rec["kernels"][k] = e
        if ("stage_band" in k or "stage_up" in k or "tile_stat" in k) and e.get("hbm_bytes") is not None:
            # the final stage may be several kernels (hybrid: statistic + band kernel + full-table kernel)
            rec["final_stage_bytes_per_launch"] = rec.get("final_stage_bytes_per_launch", 0.0) + e["hbm_bytes"]
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "profiles/hbm_traffic.json")
