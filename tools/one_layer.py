"""Run one conv layer (fwd, dgrad, wgrad) a few times: a target for rocprofv3 --pmc.  usage: one_layer.py <kbench filter>"""
import subprocess, sys, os
os.environ["KONLY"] = "1"
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "cgen.up5"]
exec(open(os.path.join(os.path.dirname(__file__), "kbench.py")).read())
