import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f,"ERR",e); continue
    print(f, "value %.2f ms/step %.0f" % (d["value"], d["ms_per_step"]), d.get("device_gate"))
    print("  ", " | ".join("%s n%d %.1fms busy %.1fs" % (k["kernel"],k["launches"],k["avg_launch_ms"],k["busy_ms"]/1e3) for k in d["kernels"]))
    print("  ", d["host_seconds"], "r2truth %.5f" % d["dosage_r2_vs_truth_sample0"])
