import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "frames/s", round(d["ms_per_step"], 4), "ms/step", "rows", d.get("rows_emitted_rank0"))
