"""Regenerates tests/golden/*.{f32,u32} from the REFERENCE's own headers.

Runs only in the authoring container (needs /root/reference): builds oracle/_ref/ref_golden with the recipe in
oracle/Makefile (the reference headers are compiled from where they lie; nothing is copied) and runs it with
this directory as the output.  The fixtures are data: inputs + the reference's outputs.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(HERE, "..", "..", "oracle")

if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference (authoring container only)")
    subprocess.check_call(["make", "-C", ORACLE, "ref"])
    subprocess.check_call([os.path.join(ORACLE, "_ref", "ref_golden"), HERE])
