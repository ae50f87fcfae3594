#!/usr/bin/env python3
"""Summarise per-kernel resources and instruction mix from hipcc -S output."""
import re, sys
for f in sys.argv[1:]:
    s = open(f).read()
    res = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', s, re.S):
        res[m.group(1)] = m.group(2)
    for name, meta in res.items():
        b = re.search(r'^' + re.escape(name) + r':[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
        body = b.group(1) if b else ''
        g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', meta) or [None, '?'])[1]
        cnt = lambda p: len(re.findall(p, body))
        short = re.sub(r'^_ZN4trag', '', name)[:60]
        print(f"{short:60s} vgpr={g('next_free_vgpr')} sgpr={g('next_free_sgpr')} lds={g('group_segment_fixed_size')} "
              f"scratch={g('private_segment_fixed_size')} fma={cnt(r'v_(pk_)?fma|v_mac|v_fmac')} mul={cnt(r'v_(pk_)?mul_f32')} "
              f"add={cnt(r'v_(pk_)?add_f32')} gld4={cnt(r'global_load_dwordx4')} nt={cnt(r'global_load_dwordx4.* nt')} "
              f"sld={cnt(r's_load_dword')} dsr128={cnt(r'ds_read_b128')} dsw128={cnt(r'ds_write_b128')} bperm={cnt(r'ds_bpermute|ds_swizzle')} dpp={cnt(r'dpp')}")
