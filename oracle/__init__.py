"""CPU oracle of Origami's dewarp hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (origami_amd/) never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path, and its
third-party numerics (cv2, shapely/GEOS) cannot be executed in the build container; see DESIGN.md.
"""
