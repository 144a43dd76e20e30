#!/usr/bin/env python3
"""Collects the case files (input.nml) of every example the reference ships into tests/golden/examples.json -- input DATA for
tests/test_gpu_examples.py, which runs each case (grid shrunk) through the device path. Runs only in the build container.

Usage:  python tests/golden/gen_examples.py"""
import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
EX = "/root/reference/examples"
out = {}
for f in sorted(glob.glob(os.path.join(EX, "*", "*", "input.nml"))):
    out[os.path.relpath(os.path.dirname(f), EX)] = open(f).read()
json.dump(out, open(os.path.join(HERE, "examples.json"), "w"), indent=0, sort_keys=True)
print(len(out), "examples ->", os.path.join(HERE, "examples.json"))
