"""Static check of wino44_conv_kernel's machine code (csrc/vv_wino44.hip): the filter taps are loaded by inline asm and waited for by
hand-counted s_waitcnt, so the compiler believes a tap register is valid from the load on -- a spill or a copy of such a register
before its wait would move garbage.  This script compiles the file to assembly and, for every asm tap load (buffer_load_dwordx2
inside #ASMSTART/#ASMEND), follows the straight-line code until the first instruction that names one of the two destination
registers: it must be a v_mfma preceded (since the load) by an asm s_waitcnt vmcnt.   python tools/check_w44_asm.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'vec_vad_amd', 'csrc', 'vv_wino44.hip')


def main():
    out = '/tmp/vv_wino44_check.s'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-inline-asm', '-S',
                           '--cuda-device-only', '-o', out, SRC] + os.environ.get('VV_HIPCC_EXTRA', '').split())
    lines = open(out).read().split('\n')
    bad = 0
    nloads = 0
    in_asm = False
    for i, ln in enumerate(lines):
        if '#ASMSTART' in ln:
            in_asm = True
            continue
        if '#ASMEND' in ln:
            in_asm = False
            continue
        m = re.search(r'buffer_load_dwordx2 v\[(\d+):(\d+)\]', ln)
        if not (in_asm and m):
            continue
        nloads += 1
        regs = {int(m.group(1)), int(m.group(2))}
        waited = False
        inasm2 = False
        for j in range(i + 1, min(i + 4000, len(lines))):
            l2 = lines[j]
            if '#ASMSTART' in l2:
                inasm2 = True
            elif '#ASMEND' in l2:
                inasm2 = False
            if inasm2 and 's_waitcnt vmcnt' in l2:
                waited = True
            if l2.strip().startswith('s_endpgm'):
                break
            used = set()
            for a, b in re.findall(r'v\[(\d+):(\d+)\]', l2):
                used.update(range(int(a), int(b) + 1))
            for a in re.findall(r'\bv(\d+)\b', l2):
                used.add(int(a))
            if used & regs:
                if inasm2 and 'buffer_load_dwordx2' in l2:       # overwritten by the next load before any use (dead value)
                    break
                if not (l2.strip().startswith('v_mfma') and waited):
                    print('line %d: tap load into v[%s] is touched before its wait: %s' % (j + 1, m.group(0), l2.strip()))
                    bad += 1
                break
    print('%d asm tap loads checked, %d violations' % (nloads, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
