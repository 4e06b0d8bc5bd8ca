#!/usr/bin/env python3
"""Closest-hit traversal on a hierarchy that does not fit the L2 (GPU box): bunny_box with the bunny tessellated to 3.7 M triangles
(tests/scenes.py: bunny_box_subdivided), forward render at 1024 x 1024 -- the same queues as the benchmark's forward pass.
Prints one JSON line: Scene build time, records per ray, algorithmic bytes per launch, mean launch duration, SURVEY 8d fraction.
  python tools/large_scene_trace.py [levels [spp]]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import bench
    print(json.dumps(bench.large_hierarchy_leg(levels, spp)), flush=True)


if __name__ == '__main__':
    main()
