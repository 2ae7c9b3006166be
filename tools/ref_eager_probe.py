#!/usr/bin/env python3
"""Time bench.py's reference_eager_rocm leg alone (how long does stock PyTorch-ROCm eager take on the reference's op chain?)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = argparse.Namespace(size=256, capacity=16, bins=64, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 8, cpu_images=1)
t0 = time.perf_counter()
print(bench.reference_eager_rocm(a, torch.device('cuda:0')))
print(f'wall {time.perf_counter() - t0:.1f} s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
