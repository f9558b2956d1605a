import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
lo=[r for r in rows if "k_kmer_scan" in r[2]][-1][0]
rows=[r for r in rows if r[0]>=lo]
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return n[:n.index("(")] if "(" in n else n[:40]
out=[]
for i,r in enumerate(rows):
    if "copyBuffer" in r[2] or "fillBuffer" in r[2]:
        prev=next((short(rows[j][2]) for j in range(i-1,-1,-1) if "rocclr" not in rows[j][2]),"-")
        nxt=next((short(rows[j][2]) for j in range(i+1,len(rows)) if "rocclr" not in rows[j][2]),"-")
        out.append(((r[1]-r[0])/1e6,short(r[2]),prev,nxt))
for d,n,p,x in sorted(out,reverse=True)[:14]: print(f"{d:8.2f} ms {n:32s} after {p:28s} before {x}")
