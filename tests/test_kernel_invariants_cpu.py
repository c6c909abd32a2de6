"""What the headline kernel's speed rests on, checked WITHOUT a GPU: hipcc cross-compiles the INT8 gather translation unit to
gfx950 assembly (one file, ~50 s) and the resource lines of the kernels are read back.

* the headline instantiation `k_embed_wave<INT8, __half, 768, 3, FIXED_POS, !PARTIAL, HIOCC>` keeps 8 waves per SIMD (<= 64 VGPRs)
  -- the high-occupancy variant of round 2 is where 3-7 % of the kernel time came from -- and parks the wave's position row
  in LDS;
* no wave-per-token kernel spills (a spill inside the per-token loop is HBM traffic: the INT4 kernel once lost 5 % to 44 B/lane);
* a token's id record arrives by scalar loads (`s_load_dwordx8`), never by a vector load + readfirstlane (a refactor once
  turned one of them into that: -5..-11 % on three formats).

A kernel edit that breaks one of these fails here, before anybody spends a GPU minute on it."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scone_amd", "csrc")
HEADLINE = "k_embed_waveILi2E6__halfLi768ELi3ELb1ELb0ELb1E"          # <SCONE_FMT_I8, __half, 768, 3, true, false, true>


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path_factory.mktemp("asm") / "scone_gather_i8.s"
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
           "scone_gather_i8.hip", "-o", str(out)]
    p = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return open(out).read()


def _kernels(asm):
    """{mangled name: (body text, {resource: value})} for every kernel of the file (label line .. next kernel's label)."""
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):[^\n]*\n; %bb\.0:", asm, flags=re.M)]
    out = {}
    for k, (pos, name) in enumerate(labels):
        chunk = asm[pos:labels[k + 1][0] if k + 1 < len(labels) else len(asm)]
        end = chunk.find(".Lfunc_end")
        res = {a: int(b) for a, b in re.findall(r"; (NumVgprs|ScratchSize|Occupancy|NumSgprs): (\d+)", chunk)}
        lds = re.search(r"; LDSByteSize: (\d+)", chunk)
        if lds:
            res["LDSByteSize"] = int(lds.group(1))
        out[name] = (chunk[:end] if end > 0 else chunk, res)
    return out


def test_headline_kernel_keeps_its_occupancy_and_its_scalar_records(asm):
    ks = _kernels(asm)
    head = [n for n in ks if HEADLINE in n]
    assert len(head) == 1, [n for n in ks if "k_embed_wave" in n][:5]
    body, res = ks[head[0]]
    assert res["ScratchSize"] == 0 and res["NumVgprs"] <= 64 and res["Occupancy"] == 8, res
    assert res["LDSByteSize"] == 4 * 6 * 64 * 4, res                   # 4 waves x 6 words x 64 lanes: the position rows (d = 768 fp16)
    # the three record loads of the loop (first token, prefetch of the next, ...) are scalar, 8 dwords at once
    assert len(re.findall(r"\bs_load_dwordx8\b", body)) >= 2, "id records must arrive by s_load_dwordx8"
    assert not re.search(r"v_readfirstlane_b32.*\n.*v_readfirstlane_b32.*\n.*v_readfirstlane_b32.*\n.*v_readfirstlane_b32.*\n.*v_readfirstlane_b32", body), \
        "a run of v_readfirstlane: a record is being loaded by vector loads"
    # the output leaves as streaming stores: one dwordx4 + one dwordx2 per lane and token
    assert re.search(r"global_store_dwordx4 .* nt", body) and re.search(r"global_store_dwordx2 .* nt", body)


def test_no_wave_kernel_of_the_int8_unit_spills(asm):
    ks = _kernels(asm)
    wave = {n: r for n, (_, r) in ks.items() if any(k in n for k in ("k_embed_wave", "k_embed_fused", "k_embed_csr_wave"))}
    assert len(wave) >= 20, len(wave)                                  # every (dim, out dtype, max_n, variant) instantiation
    bad = {n[:90]: r for n, r in wave.items() if r.get("ScratchSize", 1) != 0}
    assert not bad, bad
