#!/usr/bin/env python3
"""Where the command-line compressor spends its wall time: 1.5 Gbases of the bench recipe as FASTQ (host generator), then
`colord_hip compress-ont -v` twice (the -v phase stamps: parse + upload + pass 1, counting, references, pass 2, header stream)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd import ontsim
t = ontsim.ReadTable(seed=31, genome_len=90_000_000, target_bases=1_500_000_000)
with tempfile.TemporaryDirectory() as tmp:
    fq = os.path.join(tmp, "in.fastq"); t0 = time.time(); nb = ontsim.write_fastq(t, fq); print("fastq", nb, "bases", round(time.time() - t0, 1), "s", os.path.getsize(fq), "bytes", flush=True)
    for k in range(2):
        t0 = time.time(); r = subprocess.run([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colord_amd", "colord_hip"), "compress-ont", "-v", fq, os.path.join(tmp, "a.colord")], capture_output=True, text=True); print("run", k, round(time.time() - t0, 2), "s"); print(r.stderr[-900:])
