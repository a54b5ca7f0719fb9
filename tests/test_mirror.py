"""The host reader of the integrator-state mirror (csrc/cnf_mirror.h: tagged 8-byte granules) against a hostile writer
thread, on the CPU: no snapshot torn between two launches is ever accepted."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mirror_reader_never_accepts_a_torn_state(tmp_path):
    exe = str(tmp_path / "mirror_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "support", "mirror_test.cpp")],
                   check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(zip(r.stdout.split()[0::2], r.stdout.split()[1::2]))
    assert int(out["bad"]) == 0 and int(out["accepted"]) > 0 and int(out["final"]) == 1
