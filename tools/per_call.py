import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "resid_minmax" in n or "cgate" in n or "bproj" in n or "scan_pair" in n or "enc_p" in n or "dec_p" in n:
        seq[n[:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in seq.items():
    # group by position within the forward (3 layers)
    k = 3 if len(v) % 3 == 0 and "enc" not in n and "dec" not in n else 1
    print(n, " ".join(f"pos{j}: {sum(v[j::k])/len(v[j::k]):.1f}" for j in range(k)), "n=", len(v))
# gaps between consecutive kernels
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
print("median gap us:", sorted(gaps)[len(gaps)//2], "mean:", sum(g for g in gaps if g < 50)/len(gaps))
