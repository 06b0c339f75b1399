import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.3f  stages %s  setup %.2fs" % (d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_last_step"].items()}, d["setup_s_untimed"]))
for k, v in d["kernels"].items():
    print("  %-32s %8.3f ms  %5.1f launches" % (k, v["ms_per_step"], v["launches_per_step"]))
r = d["roofline"]
print("roofline", r["kernel"], "%.2f %s frac %.3f" % (r["achieved"], r["unit"], r["frac"]))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_proof_identical"])
