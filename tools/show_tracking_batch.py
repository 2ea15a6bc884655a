import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d.get("mode"))
for th,v in d["by_th"].items():
    print(th, round(v["value"]), round(v["ms_per_step"],2), {k:round(x,2) for k,x in v["ms_per_step_by_part"].items()})
    print("   ", {k.replace("tracked_batch.",""):round(x,2) for k,x in v["inside_the_library"].items() if x is not None})
