"""profiles/rNN_pmc_<workload>.json from a tools/pmc_summary.py summary: per launch of the closest-hit and any-hit
kernel families — HBM-side bytes (read by bench.py for roofline.traffic), L2 read requests, L1 (TCP) accesses, TA busy.
usage: pmc_family.py <summary.json> <workload> <source-note>"""
import json, sys
d = json.load(open(sys.argv[1]))
out = {"workload": sys.argv[2], "source": sys.argv[3]}


def per_launch(ks, counter, scale=1.0):
    n = sum(d[k][counter]["dispatches"] for k in ks if k in d and counter in d[k])
    return (sum(d[k][counter]["sum"] for k in ks if k in d and counter in d[k]) * scale / n, n) if n else (None, 0)


for fam in ("closest", "any"):
    ks = [f"k_trace_{fam}_pt", f"k_trace_{fam}_packet"]
    fetch, n = per_launch(ks, "FETCH_SIZE", 1024.0)
    write, _ = per_launch(ks, "WRITE_SIZE", 1024.0)
    o = {"kernels": ks, "dispatches": n, "fetch_size_bytes_per_launch": fetch, "write_size_bytes_per_launch": write}
    for name, counter, scale in (("tcp_accesses_per_launch", "TCP_TOTAL_ACCESSES_sum", 1.0), ("tcp_cache_accesses_per_launch", "TCP_TOTAL_CACHE_ACCESSES_sum", 1.0),
                                 ("l2_read_req_per_launch", "TCP_TCC_READ_REQ_sum", 1.0), ("tcp_pending_stall_cycles_per_launch", "TCP_PENDING_STALL_CYCLES_sum", 1.0),
                                 ("ta_busy_cycles_per_launch", "TA_TA_BUSY_sum", 1.0), ("ta_flat_read_wavefronts_per_launch", "TA_FLAT_READ_WAVEFRONTS_sum", 1.0),
                                 ("ta_addr_stalled_by_tc_per_launch", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", 1.0), ("ta_data_stalled_by_tc_per_launch", "TA_DATA_STALLED_BY_TC_CYCLES_sum", 1.0),
                                 ("ea_rdreq_per_launch", "TCC_EA0_RDREQ_sum", 1.0), ("ea_rdreq_32b_per_launch", "TCC_EA0_RDREQ_32B_sum", 1.0), ("ea_rdreq_dram_per_launch", "TCC_EA0_RDREQ_DRAM_sum", 1.0),
                                 ("tcc_hit_per_launch", "TCC_HIT_sum", 1.0), ("tcc_miss_per_launch", "TCC_MISS_sum", 1.0), ("grbm_gui_active_per_launch", "GRBM_GUI_ACTIVE", 1.0)):
        v, _ = per_launch(ks, counter, scale)
        if v is not None:
            o[name] = v
    thr, _ = per_launch(ks, "SQ_THREAD_CYCLES_VALU")
    act, _ = per_launch(ks, "SQ_ACTIVE_INST_VALU")
    if thr and act:
        o["valu_lane_utilisation"] = thr / (64.0 * act)
    if "l2_read_req_per_launch" in o:
        o["l2_read_bytes_per_launch"] = o["l2_read_req_per_launch"] * 64.0  # TCP->TCC read requests are 64-byte
    out[fam] = o
out["note"] = ("per kernel family, averaged over all its launches of the run.  FETCH_SIZE / WRITE_SIZE are the counters as reported (KiB -> bytes), uncorrected: "
               "on gfx950 FETCH_SIZE counts a wide coalesced streaming read at 1/2 of its bytes (MI355X_MICROARCH.md) but a per-lane 64-byte gather at its full size "
               "(calibrated: profiles/r02_gather_bench.json, fetch_size_calibration: 65.5 counted bytes per 64-byte record of a 1 GB table); bench.py adds the missing "
               "half of the streamed ray records.  FETCH_SIZE counts Infinity-Cache hits too: an upper bound on DRAM traffic")
json.dump(out, sys.stdout, indent=1)
