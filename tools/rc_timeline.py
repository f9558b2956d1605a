import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if any(x in n for x in ("k_range_code", "k_dna_evolve", "k_evolve_small", "k_dna_walk<false>", "k_qual_symbols", "k_emit_write_wave", "k_long_apply<8>")):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n.replace("(anonymous namespace)::","").split("(")[0][-20:],r["Queue_Id"],r.get("Grid_Size","?")))
rows.sort()
t0=rows[0][0]
for s,e,n,q,g in rows[-60:]:
    print("%9.1f %9.1f  %7.1f ms  q%-3s grid %-6s %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,q,g,n))
