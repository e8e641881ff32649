import csv, json, sys
d, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f"{d}/t_kernel_stats.csv")))
try:
    j = json.load(open(f"{d}.json"))
    print("==", tag, "tok/s", j["value"], "ms/step", j["ms_per_step"])
except Exception as e:
    print("==", tag, e)
for r in rows:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    if any(k in n for k in ("gemv", "softmax", "quant_pack_k")):
        print(f"{int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us  {n[:90]}")
