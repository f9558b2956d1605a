#!/usr/bin/env python3
"""`colord_hip --stream-input` against the default (whole input resident) on a synthetic ONT FASTQ: wall time, phases, peak device memory
(sampled from hipMemGetInfo through rocm-smi every 0.2 s) and the archives' streams.  Usage: tools/stream_input_check.py [bases] [chunk_bases]"""
import hashlib, os, re, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colord_amd import ontsim
sys.path.insert(0, os.path.join(ROOT, "tests"))
bases = float(sys.argv[1]) if len(sys.argv) > 1 else 5e9
chunk = sys.argv[2] if len(sys.argv) > 2 else "500000000"
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")
table = ontsim.ReadTable(seed=103, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=int(bases))
def vram_used():
    r = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    m = re.search(r"VRAM Total Used Memory \(B\): (\d+)", r)
    return int(m.group(1)) if m else 0
with tempfile.TemporaryDirectory() as tmp:
    fq = os.path.join(tmp, "in.fastq")
    t0 = time.time(); n = ontsim.write_fastq(table, fq); print(f"{n} bases, {os.path.getsize(fq)} bytes written in {time.time() - t0:.1f} s", flush=True)
    digests = {}
    for name, extra, env in [("resident, all announced", [], {"COLORD_HIP_ANNOUNCE_WINDOW": "0"}), ("resident", [], {}), ("stream-input", ["--stream-input"], {})] * 2:
        arc = os.path.join(tmp, name.split(",")[0] + ".colord")
        peak = [0]; stop = [False]
        def watch():
            while not stop[0]:
                peak[0] = max(peak[0], vram_used()); time.sleep(0.2)
        th = threading.Thread(target=watch); th.start()
        t0 = time.time()
        r = subprocess.run([CLI, "compress-ont", "-v", "-k", "25", "-a", "22", "--part-symbols", "65536", "--chunk-bases", chunk] + extra + [fq, arc], capture_output=True, text=True, env=dict(os.environ, **env))
        dt = time.time() - t0
        stop[0] = True; th.join()
        if r.returncode != 0: print(name, "FAILED", r.stderr[-800:]); continue
        phases = "; ".join(l.strip() for l in r.stderr.splitlines() if l.startswith("["))
        from colord_amd import archive as AR
        a = AR.read_archive(arc)
        digests[name] = {s: hashlib.sha256(b"".join(p for _, p in a[s].parts)).hexdigest()[:12] for s in a if s != "info"}
        print(f"{name:24s}: {dt:6.2f} s = {n / dt / 1e9:.3f} Gbases/s; peak device memory {peak[0] / 1e9:6.1f} GB; archive {os.path.getsize(arc)} B; {phases}", flush=True)
    print("streams equal:", digests.get("resident") == digests.get("stream-input"), digests.get("stream-input"))
