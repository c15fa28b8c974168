import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    try:
        r = json.loads(l)
    except Exception:
        continue
    d[r["lib"]].append((r["ms_per_sweep"], r["leap_us"]))
for k, v in d.items():
    print(k, [x[1] for x in v], round(sum(x[0] for x in v) / len(v), 5))
