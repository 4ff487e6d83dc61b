#!/usr/bin/env python3
"""Static picture of a kernel of a model code object: basic blocks of the disassembly with their
instruction mix by issue class, loops (back edges) and what sits in them (barriers, LDS / global
reads, divisions).  No GPU needed.
usage: python tools/kernel_blocks.py <code object .hsaco> [kernel name, default sdp_sweep_col] [--dump]
Issue classes (profiles/r03_ubench_valu_rate.txt, two or more waves per SIMD):
  f64    fp64 arithmetic / compare / convert / min / max                      ~4.3 clk per wave64 instruction
  vop3   32-bit work in the VOP3 encoding (v_med3, v_lshl_add, v_cndmask with an SGPR mask, v_mul_lo, packed fp32 ..)  ~4.3
  v32    plain 32-bit VOP1 / VOP2 (v_mov_b32, v_add_u32, v_and_b32, v_or_b32, v_mul_f32, v_add_f32)  ~2.4
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
FAST32 = ('v_mov_b32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mul_f32', 'v_add_f32',
          'v_sub_f32', 'v_subrev_f32', 'v_not_b32', 'v_add_co_u32', 'v_addc_co_u32', 'v_sub_co_u32', 'v_subb_co_u32',
          'v_lshlrev_b32_e32', 'v_ashrrev_i32_e32', 'v_lshrrev_b32_e32')


def disassemble(path, kernel):
    with tempfile.TemporaryDirectory() as tmp:
        elf = os.path.join(tmp, 'k.elf')
        subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + path,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf])
        text = subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', elf]).decode()
    lines = text.splitlines()
    start = [i for i, l in enumerate(lines) if '<{}>:'.format(kernel) in l][0]
    base = int(lines[start].split()[0], 16)
    ins = []
    for l in lines[start + 1:]:
        if re.match(r'^[0-9a-f]+ <', l):
            break
        m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)', l)
        if m:
            tgt = re.search(r'<{}\+0x([0-9a-f]+)>'.format(re.escape(kernel)), l)
            ins.append(dict(off=int(m.group(3), 16) - base, op=m.group(1), args=m.group(2),
                            target=int(tgt.group(1), 16) if tgt else None))
    return ins


def issue_class(op):
    if op.startswith(('s_barrier',)):
        return 'barrier'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_setprio'):
        return 'wait'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('v_'):
        base = op
        if 'f64' in op or 'b64' in op or 'i64' in op or 'u64' in op:
            return 'f64'
        if any(base.startswith(f) for f in FAST32) and not base.endswith('_e64'):
            return 'v32'
        return 'vop3'
    return 'other'


def blocks_of(ins):
    leaders = {0}
    for i, x in enumerate(ins):
        if x['op'].startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
            if i + 1 < len(ins):
                leaders.add(ins[i + 1]['off'])
            if x['target'] is not None:
                leaders.add(x['target'])
    out, cur = [], []
    for x in ins:
        if x['off'] in leaders and cur:
            out.append(cur)
            cur = []
        cur.append(x)
    if cur:
        out.append(cur)
    return out


def main():
    path = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 'sdp_sweep_col'
    ins = disassemble(path, kernel)
    if '--dump' in sys.argv:
        for x in ins:
            print('{:6x}  {:28s} {}'.format(x['off'], x['op'], x['args']))
        return
    blocks = blocks_of(ins)
    loops = []
    for b in blocks:
        last = b[-1]
        if last['target'] is not None and last['target'] <= last['off']:
            loops.append((last['target'], last['off']))
    print('{}: {} instructions, {} blocks, {} back edges'.format(kernel, len(ins), len(blocks), len(loops)))
    print('loops (from, to, instructions inside, mix):')
    for lo, hi in sorted(loops):
        body = [x for x in ins if lo <= x['off'] <= hi]
        mix = {}
        for x in body:
            c = issue_class(x['op'])
            mix[c] = mix.get(c, 0) + 1
        tags = []
        if any(x['op'] == 's_barrier' for x in body):
            tags.append('barrier x{}'.format(sum(x['op'] == 's_barrier' for x in body)))
        for key in ('v_div_scale_f64', 'v_rcp_iflag_f32', 'ds_read_b128', 'ds_write_b128', 'global_load_dwordx4', 'global_load_dwordx2',
                    'global_atomic', 'v_cvt_i32_f64'):
            n = sum(x['op'].startswith(key) for x in body)
            if n:
                tags.append('{} x{}'.format(key, n))
        print('  {:6x} .. {:6x}  {:5d}  {}   {}'.format(lo, hi, len(body), ' '.join('{}={}'.format(k, v) for k, v in sorted(mix.items())),
                                                     ', '.join(tags)))


if __name__ == '__main__':
    main()
