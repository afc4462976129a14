"""Compiler-dependent invariants of the hand-scheduled kernels, checked on the ASSEMBLY hipcc emits for gfx950 (no GPU needed).

ADVICE r5: wino_ring_kernel (vv_wino.hip) and conv_ring16_kernel (vv_conv_ring16.hip) keep HBM -> LDS DMA in flight across tile
boundaries and epilogues with hand-counted ``s_waitcnt vmcnt(N)``: N assumes that every wave issues AT LEAST ``STORES_MIN`` VMEM
instructions (its output stores) in every epilogue.  If a compiler change merged those stores into wider ones, a chunk's wait
would stop covering its DMA and the MFMAs would read LDS slots that have not landed -- silently wrong results that only the
bit-exact GPU tests of one toolchain would catch.  This test pins the assumption at build time: it compiles the two sources to
assembly and counts the store instructions of every instantiation."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'vec_vad_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _asm(src):
    out = os.path.join(tempfile.mkdtemp(prefix='vv_asm_'), 'k.s')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-inline-asm', '--cuda-device-only', '-S', '-o', out,
                           os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return open(out).read()


def _bodies(asm, needle):
    """{mangled name: text between the function label and its end marker} of every function whose name contains needle"""
    res = {}
    for m in re.finditer(r'^(_Z\w*%s\w*):' % needle, asm, re.M):
        end = asm.index('.Lfunc_end', m.end())
        res[m.group(1)] = asm[m.end():end]
    return res


def _const(src, name):
    m = re.search(r'constexpr int %s = (\d+);' % name, open(os.path.join(CSRC, src)).read())
    assert m, name
    return int(m.group(1))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
@pytest.mark.parametrize('src,kernel', [('vv_wino.hip', 'wino_ring_kernel'), ('vv_conv_ring16.hip', 'conv_ring16_kernel')])
def test_ring_kernels_issue_the_stores_their_vmcnt_arithmetic_counts(src, kernel):
    stores_min = _const(src, 'STORES_MIN')
    bodies = _bodies(_asm(src), kernel)
    assert bodies, 'no %s instantiation found in the assembly' % kernel
    for name, body in bodies.items():
        stores = re.findall(r'^\s*((?:buffer|global)_store_\w+)', body, re.M)
        # every output store of the epilogue is its own VMEM instruction: at least STORES_MIN of them in the kernel text (one epilogue
        # body per instantiation), and none of them a compiler-merged multi-dword store of the per-lane scalar outputs
        assert len(stores) >= stores_min, (name, len(stores), stores_min)
        # the counted waits are really in the text (a compiler that does not trust the inline asm would insert vmcnt(0) everywhere)
        waits = [int(x) for x in re.findall(r's_waitcnt vmcnt\((\d+)\)', body)]
        assert any(w >= stores_min for w in waits), (name, sorted(set(waits)))
        # the DMA form the ring depends on
        assert re.search(r'buffer_load_dwordx4 .* lds', body), name
