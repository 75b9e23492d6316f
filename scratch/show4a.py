import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d.get("configs", []):
    if c["config"].startswith("4a") or c["config"].startswith("2:"): print(c["config"][:20], round(c["kernel_ms"],3), round(c["roofline"]["frac"],3))
print("step", d["ms_per_step"], "greedy", d["greedy_end_to_end"]["wall_s"])
