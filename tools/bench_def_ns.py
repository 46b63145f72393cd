#!/usr/bin/env python3
"""De.Def.Ns.deflate (level 4) over 1024 x 256 KiB buffers on one MI355X: bench.py's def_ns leg on its own.
    python tools/bench_def_ns.py            (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import decompress_amd
import bench

args = argparse.Namespace(no_verify=False)
print(json.dumps(bench.def_ns_leg(args, decompress_amd.Engine(0), torch.device("cuda", 0))))
