"""Instruction mix of selected kernels from a hipcc -S listing (tools aid; not part of the product).
usage: python tools/isa_mix.py /tmp/eng.s <substring> [<substring> ...]"""
import collections
import re
import sys

KEYS = ['v_mfma_f32_16x16x4_f32', 'v_mfma_f32_16x16x32_f16', 'v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32', 'v_fma_f32',
        'v_fmac_f32_e32', 'v_mul_f32_e32', 'v_add_f32_e32', 'v_max_f32_e32', 'v_pk_max_f32', 'ds_read_b128', 'ds_read_b64',
        'ds_read_b32', 'ds_write_b128', 'global_load_dwordx4', 'global_store_dwordx4', 'scratch_load_dwordx4',
        'scratch_store_dwordx4', 'scratch_load_dword', 'scratch_store_dword', 's_waitcnt', 's_barrier', 'v_mov_b32_e32',
        'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 's_nop', 'v_cvt_f16_f32_e32', 'v_cvt_pk_f16_f32']


def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for f in re.split(r'\n(?=_ZN4fear[^\n]*:)', txt):
        name = f.split(':')[0]
        if not name.startswith('_ZN4fear') or not any(p in name for p in pats):
            continue
        body = f.split('.Lfunc_end')[0]
        c = collections.Counter()
        for line in body.split('\n'):
            m = re.match(r'\s+([a-z_0-9]+)\s', line)
            if m:
                c[m.group(1)] += 1
        print(name[:110])
        print('  ', {k: c[k] for k in KEYS if c[k]}, 'total', sum(c.values()))


if __name__ == '__main__':
    main()
