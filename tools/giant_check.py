#!/usr/bin/env python3
"""How many giant gaps a synthetic set has and whether the tile-job aligner (align_giant.hpp) finished them: the command-line compressor
with COLORD_HIP_GAP_DEBUG on a FASTQ of the bench recipe, its archive against the one written with the giants sent to the wave kernel."""
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd import ontsim
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")
bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
table = ontsim.ReadTable(seed=29, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=bases)
with tempfile.TemporaryDirectory() as tmp:
    fq = os.path.join(tmp, "in.fastq"); ontsim.write_fastq(table, fq)
    out = {}
    for name, env in (("giants", {"COLORD_HIP_GAP_DEBUG": "1"}), ("wave", {"COLORD_HIP_NO_TEAM_ALIGN": "1"})):
        arc = os.path.join(tmp, name + ".colord")
        t = time.time()
        r = subprocess.run([CLI, "compress-ont", "-k", "25", "-a", "22", "--chunk-bases", str(max(6e7, bases / 5)), fq, arc], capture_output=True, text=True, env=dict(os.environ, **env))
        print(name, "rc", r.returncode, "%.1f s" % (time.time() - t), flush=True)
        for l in r.stderr.splitlines():
            if "giant" in l or r.returncode:
                print("   ", l)
        out[name] = hashlib.sha256(open(arc, "rb").read()[:-200]).hexdigest() if r.returncode == 0 else None
    print("archives equal (but for the tail with the info stream):", out["giants"] == out["wave"] and out["giants"] is not None)
