#!/usr/bin/env python3
"""The discriminating experiment on DESIGN.md §4.6's "atomic in the walk" fault (round 3).

Round 2 showed: the round-1 decoder (commit 465d566) with `atomicAdd(rst_check, 1)` INSIDE the divergent decode loop
of the write pass stores ~510 000 wrong pixels, differently on every run; every variant that waits for the atomic, or
issues it after the loop, is correct.  Two explanations fit that: (a) a race of the kernel's own protocol — two walks
writing one coefficient slot — that only shows when the timing changes; (b) something below the source level.

This tool rebuilds that decoder so that ONE decode call runs, back to back on the same settled chunk states:
  pass A  the write pass with the atomic inside the loop (the faulty variant C), and
  pass B  the write pass as committed (count in a register, one atomic after the loop),
each into its own coefficient / DC buffers, and then, on the device, compares A with B coefficient by coefficient:
A = 0 where B stored a value, A != 0 where B stored nothing, different values; which chunks the damaged coefficients
belong to (pass B — a COPY of the kernel under another name — tags every coefficient with the chunk that wrote it: every
coefficient has ONE owner in the settled parse both passes walk), how those chunks sit in their wavefronts (64
consecutive chunks), and whether those wavefronts hold a lane that took a restart, i.e. executed the atomic.
Pass A is round 2's variant C to the letter — its kernel text is not touched by the probe (a first form of this
tool that gave pass A a template flag and one store after its loop did not show the fault at all: 0 differences in
86 M coefficients, profiles/r03/i_jpeg_owner_probe.log — the fault lives in one particular code shape).

  python tools/jpeg_atomic_owner_probe.py            # CPU box: builds origami_amd/lib/libogd_varP.so
  OGD_JPEG_PROBE=1 OGDEWARP_LIB=.../libogd_varP.so python -m pytest tests/test_gpu_jpeg_decode.py -k long_colour_streams -s
"""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "origami_amd" / "csrc"
OUT = ROOT / "origami_amd" / "lib"
TMP = Path("/tmp/ogd_jpeg_variants")


def patch(src):
	def rep(a, b):
		nonlocal src
		assert src.count(a) >= 1, a
		src = src.replace(a, b, 1)

	head = "template <bool WRITE, bool MULTI, bool RST>\n__global__ __launch_bounds__(256) void decode_chunks_kernel("
	tail = "// DC prediction in a scan with restart intervals:"
	k0, k1 = src.index(head), src.index(tail)
	kernel = src[k0:k1]
	# pass B: a COPY of the kernel as committed (count in a register, one atomic after the loop) under another name,
	# which also tags every coefficient with the chunk that wrote it and notes which walks took a restart
	clean = kernel.replace("void decode_chunks_kernel(", "void decode_chunks_clean(")
	clean = clean.replace("int restart, int* __restrict__ rst_check) {",
		"int restart, int* __restrict__ rst_check, int32_t* __restrict__ probe_tag,\n"
		"                                                            uint8_t* __restrict__ probe_took) {")
	clean = clean.replace("          coef[((size_t)(k >> 3) * ntot + slot) * 8 + (k & 7)] = (int16_t)v;",
		"          coef[((size_t)(k >> 3) * ntot + slot) * 8 + (k & 7)] = (int16_t)v;\n"
		"          probe_tag[((size_t)(k >> 3) * ntot + slot) * 8 + (k & 7)] = i;")
	clean = clean.replace("    if (taken) atomicAdd(rst_check, taken);", "    if (taken) atomicAdd(rst_check, taken);\n    probe_took[i] = taken ? 1 : 0;")
	assert "probe_tag[" in clean and "probe_took[i]" in clean
	# pass A: variant C of round 2, nothing else touched — the atomic inside the loop, none after it
	faulty = kernel.replace("            if (WRITE) {\n              ++taken;", "            if (WRITE) {\n              atomicAdd(rst_check, 1);\n              ++taken;")
	faulty = faulty.replace("    if (taken) atomicAdd(rst_check, taken);", "")
	assert faulty != kernel
	src = src[:k0] + faulty + clean + PROBE_KERNELS + "\n" + src[k1:]
	# host: after the (faulty) write pass, run pass B and compare when OGD_JPEG_PROBE is set; go on with B's buffers
	rep("  OGD_TRY(check_launch(\"decode_chunks_kernel<write>\"));\n  // D5:", "  OGD_TRY(check_launch(\"decode_chunks_kernel<write>\"));\n" + PROBE_HOST + "  // D5:")
	return src


PROBE_KERNELS = r'''
struct ProbeOut {
  unsigned long long coef_bad, lost, foreign, wrong_value, dc_bad, is_next, is_prev, is_last_of_block;
  unsigned long long bad_chunks, bad_waves, bad_lanes_in_bad_waves, bad_waves_with_restart_lane, bad_chunks_that_took_restart,
      waves_with_restart_lane, waves;
  int samples;                 // entries of sample[]
  long long sample[32][5];     // coefficient index, A, B, owner chunk (pass B), its lane
};

__global__ void probe_coef_kernel(const int16_t* __restrict__ A, const int16_t* __restrict__ B, const int32_t* __restrict__ tag,
                                  size_t n, ProbeOut* out, unsigned* __restrict__ bad_per_chunk, size_t ntot) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int a = A[e], b = B[e];
  if (a == b) return;
  atomicAdd(&out->coef_bad, 1ull);
  if (a == 0) atomicAdd(&out->lost, 1ull);
  else if (b == 0) atomicAdd(&out->foreign, 1ull);
  else atomicAdd(&out->wrong_value, 1ull);
  const int owner = tag[e];
  if (owner >= 0) atomicAdd(bad_per_chunk + owner, 1u);
  {
    // is the wrong value the one the same walk stored NEXT (or just before) in this block?  (zig-zag index k of
    // coefficient e: e = ((k >> 3) * ntot + slot) * 8 + (k & 7))
    const size_t q = e / 8, slot = q % ntot;
    const int k = (int)(q / ntot) * 8 + (int)(e % 8);
    int next = 0, prev = 0;
    for (int k2 = k + 1; k2 < 64 && !next; ++k2) next = B[((size_t)(k2 >> 3) * ntot + slot) * 8 + (k2 & 7)];
    for (int k2 = k - 1; k2 >= 1 && !prev; --k2) prev = B[((size_t)(k2 >> 3) * ntot + slot) * 8 + (k2 & 7)];
    if (next && a == next) atomicAdd(&out->is_next, 1ull);
    if (prev && a == prev) atomicAdd(&out->is_prev, 1ull);
    if (!next) atomicAdd(&out->is_last_of_block, 1ull);
  }
  const int s = atomicAdd(&out->samples, 1);
  if (s < 32) {
    out->sample[s][0] = (long long)e;
    out->sample[s][1] = a;
    out->sample[s][2] = b;
    out->sample[s][3] = owner;
    out->sample[s][4] = owner >= 0 ? owner & 63 : -1;
  }
}

__global__ void probe_dc_kernel(const int32_t* __restrict__ A, const int32_t* __restrict__ B, size_t n, ProbeOut* out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n && A[e] != B[e]) atomicAdd(&out->dc_bad, 1ull);
}

// 64 consecutive chunks are one wavefront of the write pass: where do the damaged chunks sit, and do their wavefronts
// hold a lane that took a restart (i.e. executed the atomic inside the loop)?
__global__ void probe_chunks_kernel(const unsigned* __restrict__ bad_per_chunk, const uint8_t* __restrict__ took, int nchunk,
                                    ProbeOut* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool bad = i < nchunk && bad_per_chunk[i] != 0;
  const bool rst = i < nchunk && took[i] != 0;
  const unsigned long long bb = __ballot(bad), rb = __ballot(rst);
  if (bad) atomicAdd(&out->bad_chunks, 1ull);
  if (bad && rst) atomicAdd(&out->bad_chunks_that_took_restart, 1ull);
  if ((threadIdx.x & 63) == 0 && i < nchunk) {
    atomicAdd(&out->waves, 1ull);
    if (rb) atomicAdd(&out->waves_with_restart_lane, 1ull);
    if (bb) {
      atomicAdd(&out->bad_waves, 1ull);
      atomicAdd(&out->bad_lanes_in_bad_waves, (unsigned long long)__popcll(bb));
      if (rb) atomicAdd(&out->bad_waves_with_restart_lane, 1ull);
    }
  }
}
'''

PROBE_HOST = r'''  if (std::getenv("OGD_JPEG_PROBE") && multi && rst) {
    // dcoef / ddc now hold pass A (the write pass with the atomic inside its loop, exactly round 2's variant C)
    DevBuf bcoef, bdc, btag, btook, pout, pbad, bcheck;
    const size_t ncoef = (size_t)nblk * 64;
    OGD_TRY(bcoef.alloc(ncoef * sizeof(int16_t)));
    OGD_TRY(bdc.alloc((size_t)nblk * sizeof(int32_t)));
    OGD_TRY(btag.alloc(ncoef * sizeof(int32_t)));
    OGD_TRY(btook.alloc((size_t)nchunk));
    OGD_TRY(pout.alloc(sizeof(ProbeOut)));
    OGD_TRY(pbad.alloc((size_t)nchunk * sizeof(unsigned)));
    OGD_TRY(bcheck.alloc(16));
    OGD_HIP(hipStreamSynchronize(st));  // pass A ran alone, as in round 2
    OGD_HIP(hipMemsetAsync(pout.p, 0, sizeof(ProbeOut), st));
    OGD_HIP(hipMemsetAsync(bcoef.p, 0, ncoef * sizeof(int16_t), st));
    OGD_HIP(hipMemsetAsync(bdc.p, 0, (size_t)nblk * sizeof(int32_t), st));
    OGD_HIP(hipMemsetAsync(btag.p, 0xff, ncoef * sizeof(int32_t), st));
    OGD_HIP(hipMemsetAsync(btook.p, 0, (size_t)nchunk, st));
    OGD_HIP(hipMemsetAsync(pbad.p, 0, (size_t)nchunk * sizeof(unsigned), st));
    OGD_HIP(hipMemsetAsync(bcheck.p, 0, 16, st));
    hipLaunchKernelGGL((decode_chunks_clean<true, true, true>), cgrid, blk, 0, st, dstream.as<uint8_t>(), total_bits, nchunk, tabs,
                       dentry.as<ChunkState>(), (ChunkState*)nullptr, (uint64_t*)nullptr, (int*)nullptr, dnb.as<uint64_t>(), nmcu,
                       bcoef.as<int16_t>(), bdc.as<int32_t>(), 0, rst_bits, nrst, info.restart, bcheck.as<int>(),
                       btag.as<int32_t>(), btook.as<uint8_t>());
    OGD_TRY(check_launch("pass B"));
    hipLaunchKernelGGL(probe_coef_kernel, dim3((unsigned)((ncoef + 255) / 256)), blk, 0, st, dcoef.as<int16_t>(),
                       bcoef.as<int16_t>(), btag.as<int32_t>(), ncoef, pout.as<ProbeOut>(), pbad.as<unsigned>(), (size_t)nblk);
    hipLaunchKernelGGL(probe_dc_kernel, dim3((unsigned)((nblk + 255) / 256)), blk, 0, st, ddc.as<int32_t>(), bdc.as<int32_t>(),
                       (size_t)nblk, pout.as<ProbeOut>());
    hipLaunchKernelGGL(probe_chunks_kernel, cgrid, blk, 0, st, pbad.as<unsigned>(), btook.as<uint8_t>(), nchunk, pout.as<ProbeOut>());
    OGD_TRY(check_launch("probe kernels"));
    ProbeOut h;
    int acheck[2] = {0, 0}, bchk[2] = {0, 0};
    OGD_HIP(hipMemcpyAsync(&h, pout.p, sizeof h, hipMemcpyDeviceToHost, st));
    OGD_HIP(hipMemcpyAsync(acheck, dcheck.p, sizeof acheck, hipMemcpyDeviceToHost, st));
    OGD_HIP(hipMemcpyAsync(bchk, bcheck.p, sizeof bchk, hipMemcpyDeviceToHost, st));
    // go on with pass B's buffers: the decode call returns the right pixels
    OGD_HIP(hipMemcpyAsync(dcoef.p, bcoef.p, ncoef * sizeof(int16_t), hipMemcpyDeviceToDevice, st));
    OGD_HIP(hipMemcpyAsync(ddc.p, bdc.p, (size_t)nblk * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    OGD_HIP(hipMemcpyAsync(dcheck.p, bcheck.p, 8, hipMemcpyDeviceToDevice, st));
    OGD_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "PROBE %dx%d: %d chunks = %llu wavefronts of the write pass, %lld blocks, %d restart intervals\n", info.W, info.H,
            nchunk, h.waves, (long long)nblk, nrst);
    fprintf(stderr, "PROBE restarts counted: pass A (atomic in the loop) %d, pass B (one atomic per walk) %d of %d; misplaced A %d B %d\n",
            acheck[0], bchk[0], nrst, acheck[1], bchk[1]);
    fprintf(stderr, "PROBE coefficients A != B: %llu of %zu  (A = 0 where B stored a value: %llu, A != 0 where B stored nothing: %llu, both non-zero: %llu); DC differences: %llu of %lld\n",
            h.coef_bad, ncoef, h.lost, h.foreign, h.wrong_value, h.dc_bad, (long long)nblk);
    fprintf(stderr, "PROBE of the %llu damaged coefficients, A holds the value pass B stored NEXT in the same block: %llu; the one stored just BEFORE: %llu; %llu are the last coefficient of their block\n",
            h.coef_bad, h.is_next, h.is_prev, h.is_last_of_block);
    fprintf(stderr, "PROBE damaged coefficients belong to %llu chunks in %llu wavefronts (%.1f damaged lanes per such wavefront)\n",
            h.bad_chunks, h.bad_waves, h.bad_waves ? (double)h.bad_lanes_in_bad_waves / (double)h.bad_waves : 0.0);
    fprintf(stderr, "PROBE wavefronts with a lane that took a restart (= executed the in-loop atomic): %llu of %llu; damaged wavefronts among them: %llu of %llu damaged; damaged chunks that took a restart themselves: %llu of %llu\n",
            h.waves_with_restart_lane, h.waves, h.bad_waves_with_restart_lane, h.bad_waves, h.bad_chunks_that_took_restart, h.bad_chunks);
    for (int s2 = 0; s2 < std::min(h.samples, 10); ++s2)
      fprintf(stderr, "PROBE   coef %lld: A %lld, B %lld, owner chunk %lld (lane %lld)\n", h.sample[s2][0], h.sample[s2][1],
              h.sample[s2][2], h.sample[s2][3], h.sample[s2][4]);
  }
'''


def main():
	TMP.mkdir(exist_ok=True)
	src = subprocess.run(["git", "show", "465d566:origami_amd/csrc/ogd_jpeg_dec.hip"], cwd=ROOT, check=True,
		stdout=subprocess.PIPE).stdout.decode()
	src = patch(src)
	if "#include <cstring>" not in src:
		src = src.replace("#include \"ogd_common.h\"", "#include \"ogd_common.h\"\n#include <cstring>", 1)
	# entry points the round-3 library has and that commit lacks (not used by the probe's test)
	src += '''
namespace ogd {
int jpeg_decode_gray_to(DevBuf&, const uint8_t*, size_t, int*, int*, void*) { return fail(OGD_ERR_UNSUPPORTED, "probe build"); }
}
extern "C" int ogd_jpeg_header(const uint8_t* file, size_t n, int* H, int* W, int* ncomp, int* plausible) {
  return ogd_jpeg_info(file, n, H, W, ncomp, plausible);
}
'''
	(TMP / "jdP.hip").write_text(src)
	others = [str(p) for p in (ROOT / "origami_amd" / "build").glob("*.o") if p.name != "ogd_jpeg_dec.o"]
	flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-sanitize", "-I" + str(CSRC)]
	obj = TMP / "jdP.o"
	subprocess.run(["hipcc"] + flags + ["-c", str(TMP / "jdP.hip"), "-o", str(obj)], check=True)
	subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-sanitize", "-o",
		str(OUT / "libogd_varP.so")] + others + [str(obj)], check=True)
	print("built", OUT / "libogd_varP.so")


if __name__ == "__main__":
	main()
