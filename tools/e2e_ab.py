#!/usr/bin/env python3
"""A/B of two colord_hip binaries on one FASTQ of the bench recipe (file -> archive, resident input), alternating.
Usage: tools/e2e_ab.py bases binA binB [repeats]; E2E_AB_PAUSE="0,8": seconds to wait before the runs of A / of B (the driver clears the
memory a process gave back before the next one gets it: a run that starts right after another one pays for that)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd import ontsim
bases = float(sys.argv[1]); bins = sys.argv[2:4]; rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
t = ontsim.ReadTable(seed=41, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=int(bases))
pause = [float(x) for x in os.environ.get("E2E_AB_PAUSE", "0").split(",")]
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
    fq = os.path.join(tmp, "in.fastq"); nb = ontsim.write_fastq(t, fq); os.sync()
    for r in range(rep):
        for bi, b in enumerate(bins):
            time.sleep(pause[bi % len(pause)])
            t0 = time.time()
            p = subprocess.run([b, "compress-ont", "-v", "-k", "25", "-a", "22", "--part-symbols", "65536", fq, os.path.join(tmp, "a.colord")], capture_output=True, text=True)
            dt = time.time() - t0
            ph = " | ".join(l.strip() for l in p.stderr.splitlines() if l.startswith("["))
            print(f"{os.path.basename(b)}: exit {p.returncode} {dt:.2f} s = {nb / dt / 1e9:.3f} Gbases/s :: {ph}", flush=True)
