"""Several stages over the same pages in ONE process, page round by page round.

    python -m origami_amd.batch.chain DATA_PATH --stage "utils.dewarp" --stage "export.lines --plain" [--pages-per-round 16]

In the reference's workflow every stage is a process of its own (ref: quickstart/02_cpu.sh:11-16) and every stage
that looks at pixels opens the page and remaps it again (ref: origami/batch/core/io.py:275-280, origami/core/
page.py:82-88).  Run through this command the stages share `origami_amd.page.resident`: a page is read, decoded and
uploaded by the first stage that needs its pixels and stays on the device for the others.  The stages themselves are
the ordinary commands (`origami_amd.batch.<name>`), given a listing of the round's pages instead of DATA_PATH — their
options, skip / overwrite rules, locks and runtime.json entries are what they are when run alone.  A round holds as
many pages as should stay resident together (OGD_PAGE_CACHE_MB bounds the bytes either way).

One process means one GPU and no spawned workers here (a stage's --gpus option starts workers of its own, which share
nothing): for more processes run several chains on disjoint parts of the tree (--name), as the reference runs several
stage processes on one tree."""

import importlib
import shlex
import sys
import tempfile
from pathlib import Path

import click

from .core.processor import Processor, chunks


def stage_command(name):
	"""The click command of stage module `origami_amd.batch.<name>` (e.g. "utils.dewarp", "detect.dewarp")."""
	module = importlib.import_module("origami_amd.batch." + name)
	commands = [v for v in vars(module).values() if isinstance(v, click.Command) and v.callback.__module__ == module.__name__]
	if len(commands) != 1:
		raise click.UsageError("%s: expected one stage command, found %d" % (module.__name__, len(commands)))
	return commands[0]


def pages_below(data_path, name=""):
	"""The page files a stage would walk (ref: origami/batch/core/processor.py:425-474), in its order."""
	import re
	walker = Processor({})
	root = Path(data_path)
	found = Processor._pages_below(root) if root.is_dir() else Processor._listed_pages(root)
	return [p for p in found if (not name or re.search(name, str(p))) and walker.is_image(p)]


def run_chain(data_path, stages, pages_per_round=16, name="", echo=print):
	"""stages: [(module name, [options])]; returns the number of pages walked."""
	commands = [(stage_command(n), list(opts)) for n, opts in stages]
	pages = pages_below(data_path, name)
	root = Path(data_path)
	lock_db = (root if root.is_dir() else root.parent) / "origami.lock.db"
	with tempfile.TemporaryDirectory(prefix="ogd_chain_") as tmp:
		for k, group in enumerate(chunks(pages, max(1, pages_per_round))):
			listing = Path(tmp) / ("round%06d.txt" % k)
			listing.write_text("".join("%s\n" % p for p in group))
			for command, opts in commands:
				args = [str(listing)] + opts
				if "--lock-database" not in opts and not any(o.upper() in ("NONE", "FILE") for o in opts):
					args += ["--lock-database", str(lock_db)]  # the tree's own lock database, not one beside the listing
				try:
					command.main(args=args, standalone_mode=False)
				except SystemExit as e:
					if e.code not in (0, None):
						raise
	return len(pages)


@click.command()
@click.argument("data_path", type=click.Path(exists=True), required=True)
@click.option("--stage", "stages", multiple=True, required=True,
	help='A stage and its options, e.g. "utils.dewarp --overwrite"; give the option once per stage, in order.')
@click.option("--pages-per-round", type=int, default=16, help="Pages every stage works through before the next pages are taken up.")
@click.option("--name", type=str, default="", help="Regular expression a page path must match.")
def chain(data_path, stages, pages_per_round, name):
	"""Run several stages over DATA_PATH in one process, sharing the pages they load."""
	parsed = []
	for text in stages:
		words = shlex.split(text)
		parsed.append((words[0], words[1:]))
	n = run_chain(data_path, parsed, pages_per_round, name)
	from ..page import resident
	print("%d pages; resident pages: %d hits, %d misses, %d evictions." % (n, resident.hits, resident.misses, resident.evictions))


if __name__ == "__main__":
	chain()
