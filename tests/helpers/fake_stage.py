"""A stage with the dewarp stage's artifacts but no GPU work: lets the batch runtime (queue, locks,
spawned per-GPU workers) be tested on CPU."""

import json
import os
import time
from pathlib import Path

from origami_amd.batch.core.io import Artifact, Input, Output, Stage
from origami_amd.batch.core.processor import Processor


class FakeDewarpProcessor(Processor):
	@property
	def processor_name(self):
		return "fake.dewarp"

	def artifacts(self):
		return [
			("warped", Input(Artifact.CONTOURS, Artifact.FLOW, stage=Stage.WARPED)),
			("output", Output(Artifact.DEWARPING_TRANSFORM, Artifact.CONTOURS, stage=Stage.DEWARPED)),
		]

	def process(self, page_path, warped, output):
		if page_path.stem.startswith("bad"):
			raise RuntimeError("synthetic failure")
		if page_path.stem.startswith("kill"):  # what a GPU fault / the OOM killer does to a worker: no unwinding
			import signal
			os.kill(os.getpid(), signal.SIGKILL)
		info = dict(pid=os.getpid(), rank=self._rank, device=os.environ.get("OGD_DEVICE"))
		rendezvous = os.environ.get("FAKE_STAGE_RENDEZVOUS")  # "<dir>:<n>": hold the first page until n workers run
		if rendezvous:
			root, n = rendezvous.rsplit(":", 1)
			(Path(root) / (".started.%s" % self._rank)).touch()
			deadline = time.time() + 30
			while len(list(Path(root).glob(".started.*"))) < int(n) and time.time() < deadline:
				time.sleep(0.01)
		with output.contours(copy_meta_from=warped) as zf:
			zf.writestr("who.json", json.dumps(info))
		with output.dewarping_transform() as f:
			f.write(json.dumps(info).encode("utf8"))
		return dict(worker=self._rank)
