"""NUMA placement of GPU workers.

A worker stages 48 MB up and 49 MB down per broadsheet page through pinned host memory; on a two-socket node a
worker running on the other socket sends every page across the inter-socket link twice.  A worker therefore pins
itself — and, by first touch, the pinned buffers it allocates afterwards — to the cores of the NUMA node its GPU
hangs off.  Best effort: on a machine without NUMA information (node -1, no sysfs, containers that hide it), or
when the node's cores are not in the process's affinity mask, nothing changes and the worker runs as before.

The reference has no placement of its own: its pool workers are plain forked processes (ref:
origami/batch/core/processor.py:483); it has no device whose locality could matter.
"""

import os
from pathlib import Path


def parse_cpulist(text):
	"""'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11} (the format of /sys/devices/system/node/node*/cpulist)."""
	cpus = set()
	for part in text.strip().split(","):
		part = part.strip()
		if not part:
			continue
		first, _, last = part.partition("-")
		cpus.update(range(int(first), int(last or first) + 1))
	return cpus


def node_of_pci_device(bus_id, sysfs="/sys"):
	"""NUMA node of the PCI device '0000:c1:00.0', or None if the kernel does not say (file missing, -1)."""
	try:
		node = int((Path(sysfs) / "bus" / "pci" / "devices" / bus_id.lower() / "numa_node").read_text())
	except (OSError, ValueError):
		return None
	return node if node >= 0 else None


def cpus_of_node(node, sysfs="/sys"):
	try:
		return parse_cpulist((Path(sysfs) / "devices" / "system" / "node" / ("node%d" % node) / "cpulist").read_text())
	except (OSError, ValueError):
		return set()


def bind_to_gpu_numa_node(device, sysfs="/sys", bus_id=None, setaffinity=None, getaffinity=None):
	"""Restrict the calling process to the cores of GPU `device`'s NUMA node.  Returns the node number, or None if
	the binding could not be made (the affinity is then left alone).  bus_id / setaffinity / getaffinity: injected by
	the tests (a fake sysfs tree needs no GPU)."""
	try:
		if bus_id is None:
			from . import _shim
			bus_id = _shim.device_pci_bus_id(device)
		if not bus_id:
			return None
		node = node_of_pci_device(bus_id, sysfs)
		if node is None:
			return None
		allowed = (getaffinity or (lambda: os.sched_getaffinity(0)))()
		cpus = cpus_of_node(node, sysfs) & set(allowed)
		if not cpus:
			return None
		(setaffinity or (lambda c: os.sched_setaffinity(0, c)))(cpus)
		return node
	except Exception:
		return None


def auto_workers_per_gpu(n_gpus, cpus=None, cap=8):
	"""Workers per GPU when the command line does not say: the `dewarp` stage is host-bound (5.5 ms/page with 8
	workers on one GPU against 25.8 with one), so as many as half the usable cores allow, at most
	`cap` (a GPU box admits 16 processes per GPU)."""
	if cpus is None:
		try:
			cpus = len(os.sched_getaffinity(0))
		except (AttributeError, OSError):
			cpus = os.cpu_count() or 1
	return max(1, min(cap, cpus // max(1, n_gpus) // 2))
