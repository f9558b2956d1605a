import json,sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(f,"NO JSON"); print(open(f).read()[-1500:]); continue
    d=json.loads(l[-1])
    print(f, "value=%.3f Gb/s ms/step=%.1f"%(d["value"], d["ms_per_step"]), d["config"]["sizes"].get("qual_bytes"), d["config"]["sizes"].get("qual_parts"))
    ks=d["roofline"]["kernel_ms_per_step"]; print("  kernels total %.1f ms:"%sum(ks.values()), {k:round(v,2) for k,v in list(ks.items())[:8]})
