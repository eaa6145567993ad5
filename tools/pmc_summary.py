"""Summarise tools/pmc_1x1.sh output: last 10 dispatches of the conv kernel (tiled igemm_kernel, streaming gemm1x1_kernel or wino_kernel, whichever the tuner chose). Usage: python tools/pmc_summary.py <tag> GFLOP ALGBYTES_MB"""
import csv, glob, sys
tag = sys.argv[1]; gflop = float(sys.argv[2]); mb = float(sys.argv[3])
def last(d, counter):
    f = max(glob.glob(f"gpurun_out/pmc_{tag}_{d}/**/*counter_collection.csv", recursive=True), key=__import__("os").path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ("igemm_kernel", "gemm1x1_kernel", "wino_kernel", "wino4_kernel"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-10:]
    return sum(float(r["Counter_Value"]) for r in rows) / len(rows), rows[-1]
def dur(d):
    f = max(glob.glob(f"gpurun_out/pmc_{tag}_{d}/**/*kernel_trace.csv", recursive=True), key=__import__("os").path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "gemm1x1_kernel", "wino_kernel", "wino4_kernel"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-10:]
    return sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / len(rows) / 1e3, rows[-1]
us, r = dur("a")
print(tag, "kernel", r["Kernel_Name"][:60], "grid", r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], "wg", r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "vgpr", r["VGPR_Count"])
print("  duration %.1f us -> %.1f TF" % (us, gflop * 1e3 / us))
mf, _ = last("a", "SQ_VALU_MFMA_BUSY_CYCLES"); gui, _ = last("a", "GRBM_GUI_ACTIVE")
wc, _ = last("a", "SQ_WAVE_CYCLES"); wa, _ = last("a", "SQ_WAIT_ANY"); wi, _ = last("a", "SQ_WAIT_INST_ANY"); ai, _ = last("a", "SQ_ACTIVE_INST_ANY")
cyc = gui / 8
print("  MFMA busy %.1f%% of SIMD cycles (clock %.2f GHz)" % (mf / (1024 * cyc) * 100, cyc / us / 1e3))
print("  wave cycles: wait_any %.0f%% wait_inst %.0f%% active %.0f%%" % (wa / wc * 100, wi / wc * 100, ai / wc * 100))
fe, _ = last("f", "FETCH_SIZE"); wr, _ = last("w", "WRITE_SIZE")
hbm = (2 * fe + wr) * 1024
print("  HBM traffic %.1f MB (algorithmic %.1f MB, x%.2f) -> %.2f TB/s" % (hbm / 1e6, mb, hbm / 1e6 / mb, hbm / us / 1e6))
