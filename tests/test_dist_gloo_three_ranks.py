"""The world-size-2 ``gloo`` tests once more with THREE ranks: the middle rank then has partner blocks in front of AND
behind its own block (the FILIP step scores its own block first, then the two others), which two ranks cannot show."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_steps_and_operators_on_three_ranks():
    env = dict(os.environ, PCLIP_TEST_WORLD="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_dist_gloo.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
