"""CPU suite: the argument checks of the command-line compressor that the reference makes before any work
(arg_parse.cpp:410-450,604-625) — they run before a GPU is touched."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")


def run(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True)


@pytest.mark.parametrize("args,msg", [
    (["compress-ont", "-k", "21", "in.fq", "out"], "if -k,--kmer-len is set -a,--anchor-len also must be set"),
    (["compress-ont", "-a", "18", "in.fq", "out"], "if -a,--anchor-len is set -k,--kmer-len also must be set"),
    (["compress-ont", "-k", "20", "-a", "21", "in.fq", "out"], "-a,--anchor-len must be less than or equal to -k,--kmer-len"),
    (["compress-ont", "-k", "40", "-a", "21", "in.fq", "out"], "[15, 28]"),
    (["compress-ont", "-q", "org", "-T", "7", "in.fq", "out"], "not allowed for 'org'"),
    (["compress-ont", "-q", "4-avg", "-T", "7,14", "in.fq", "out"], "expected number of quality thresholds is 3, but 2 given"),
    (["compress-ont", "-q", "4-avg", "-D", "1,2,3,4", "in.fq", "out"], "not allowed for '4-avg'"),
    (["compress-ont", "-q", "2-fix", "-D", "1", "in.fq", "out"], "expected number of quality values is 2, but 1 given"),
    (["compress-ont", "-q", "bogus", "in.fq", "out"], "unknown quality mode"),
    (["compress-ont", "-p", "fastest", "in.fq", "out"], "unknown priority"),
    (["compress-ont", "--frobnicate", "in.fq", "out"], "unknown option"),
    (["compress-ont", "in.fq"], "expected input and output paths"),
    (["compress-foo", "in.fq", "out"], "unknown mode"),
])
def test_rejected_arguments(args, msg):
    r = run(*args)
    assert r.returncode == 1 and msg in r.stderr, r.stderr


def test_usage_lists_the_sub_commands():
    r = run()
    for word in ("compress-ont", "compress-pbhifi", "compress-pbraw", "decompress", "info", "--qual-thresholds", "--identifier"):
        assert word in r.stderr
