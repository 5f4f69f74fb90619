import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("valu_busy"), d["config"]["real_text_MBps"])
print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in d["also"].items()})
print(json.dumps(d["also"].get("strong_scaling_standin"))[:200]); print(d["also"].get("one_long_stream"))
