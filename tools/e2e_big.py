#!/usr/bin/env python3
"""T_e2e at scale, once: a FASTQ of `bases` (default 20 G) of the bench recipe written by the host generator, then `colord_hip compress-ont
-k 25 -a 22 --part-symbols 65536 -v` file -> archive, resident input and --stream-input (bounded device memory).  Prints the phase stamps.
Needs 2 bytes of disk per base (checked first).  Usage: tools/e2e_big.py [bases]"""
import os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd import ontsim
bases = float(sys.argv[1]) if len(sys.argv) > 1 else 2e10
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(root, "colord_amd", "colord_hip")
base_dir = os.environ.get("TMPDIR", "/tmp")
free = shutil.disk_usage(base_dir).free
print(f"{base_dir}: {free / 1e9:.0f} GB free; need {2.6 * bases / 1e9:.0f} GB", flush=True)
if free < 2.6 * bases:
    bases = max(1e9, free / 2.6 * 0.9)
    print(f"reduced to {bases / 1e9:.1f} Gbases", flush=True)
if os.environ.get("E2E_PARENT_GPU"):                 # this process holds an (idle) HIP context while the command line runs, as bench.py does
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import torch
    from colord_amd.device import Context
    ctxs = [Context(0) for _ in range(int(os.environ["E2E_PARENT_GPU"]))]
    x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
    codes, off, _ = ontsim.device_reads(ontsim.ReadTable(seed=3, genome_len=1_000_000, target_bases=20_000_000), ctxs[0].device, 0, 100, with_quals=False)
    r = ctxs[0].pack_reads(codes, off); km = ctxs[0].kmer_scan(r, 25, 12); torch.cuda.synchronize()
    r.free(); del km, codes, off, x
    for c in ctxs:
        c.close()
    torch.cuda.empty_cache()
    nq = int(os.environ.get("E2E_PARENT_QUEUES", "0"))            # ... and that many of its own streams have seen work (hardware queues made, now idle)
    keep = [torch.cuda.Stream() for _ in range(nq)]
    for st in keep:
        with torch.cuda.stream(st):
            torch.zeros(1024, device="cuda").add_(1)
    hold = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda").fill_(1) for _ in range(int(os.environ.get("E2E_PARENT_HOLD_GB", "0")))]   # ... and it keeps that many GiB of HBM
    torch.cuda.synchronize()
    print(f"parent: holds {len(hold)} GiB of device memory", flush=True)
    print(f"parent: HIP context alive, idle ({len(ctxs)} contexts made and closed, {nq} streams used once and kept)", flush=True)
t = ontsim.ReadTable(seed=41, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=int(bases))
with tempfile.TemporaryDirectory(dir=base_dir) as tmp:
    fq = os.path.join(tmp, "in.fastq")
    t0 = time.time(); nb = ontsim.write_fastq(t, fq); os.sync()
    print(f"fastq: {nb} bases, {os.path.getsize(fq)} bytes, written in {time.time() - t0:.1f} s", flush=True)
    if os.environ.get("E2E_TINY_FIRST"):                  # a throw-away run of the command on a small sample first (is it the first GPU process that is slow, whatever its size?)
        tb = float(os.environ["E2E_TINY_FIRST"])
        tfq = os.path.join(tmp, "tiny.fastq")
        ontsim.write_fastq(ontsim.ReadTable(seed=42, genome_len=max(1_000_000, int(tb / 16.7)), target_bases=int(tb)), tfq)
        t0 = time.time()
        r = subprocess.run([cli, "compress-ont", "-k", "25", "-a", "22", "--part-symbols", "65536", tfq, os.path.join(tmp, "tiny.colord")], capture_output=True, text=True)
        print(f"tiny first run ({tb / 1e9:.2f} Gbases): exit {r.returncode}, {time.time() - t0:.2f} s", flush=True)
    runs = [("resident", [], {}), ("stream-input", ["--stream-input"], {})]
    if os.environ.get("E2E_RUNS"):                        # e.g. E2E_RUNS="pread:COLORD_HIP_COPY_PREAD=1;t64:--parse-threads=64": name:ENV=v,--flag=v,...
        runs = []
        for spec in os.environ["E2E_RUNS"].split(";"):
            name, _, rest = spec.partition(":")
            items = [x for x in rest.split(",") if x]
            extra = [y for x in items if x.startswith("--") for y in x.split("=", 1)]
            runs.append((name, extra, dict(x.split("=", 1) for x in items if not x.startswith("--"))))
    for name, extra, env in runs:
        time.sleep(8.0)
        t0 = time.time()
        r = subprocess.run([cli, "compress-ont", "-v", "-k", "25", "-a", "22", "--part-symbols", "65536"] + extra + [fq, os.path.join(tmp, f"{name}.colord")], capture_output=True, text=True, env=dict(os.environ, **env))
        dt = time.time() - t0
        print(f"{name}: exit {r.returncode}, {dt:.2f} s = {nb / dt / 1e9:.3f} Gbases/s, archive {os.path.getsize(os.path.join(tmp, name + '.colord')) if r.returncode == 0 else 0} bytes", flush=True)
        if os.path.exists(os.path.join(tmp, name + ".colord")):
            os.remove(os.path.join(tmp, name + ".colord"))          # (every run writes a NEW file: replacing one of 8 GB made close() wait 0.8 s for its blocks — ext4's replace-via-truncate rule)
        print("\n".join(l for l in r.stderr.splitlines() if l.startswith(("[", "colord_hip", "# pass")))[-1500:], flush=True)
