"""Static instruction mix of the kernels in a hipcc -save-temps .s file (per thread, loops counted once)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "pass_kernel"
for m in re.finditer(r'\n(_Z\w*' + pat + r'\w*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S):
    b = m.group(2)
    lines = [l.strip() for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    lines = [l for l in lines if not re.match(r'^[\w.$]+:', l)]
    c = lambda p: sum(1 for l in lines if re.match(p, l))
    print(m.group(1)[-60:], '| total', len(lines), 'valu', c(r'v_'), 'salu', c(r's_'), 'ds_read', c(r'ds_read'), 'ds_write', c(r'ds_write'),
          'vmem_ld', c(r'(global|buffer|flat)_load'), 'vmem_st', c(r'(global|buffer|flat)_store'), 'fma', c(r'v_fma'), 'cndmask', c(r'v_cndmask'),
          'rcp', c(r'v_rcp'), 'barrier', c(r's_barrier'), 'waitcnt', c(r's_waitcnt'), 'scratch', c(r'scratch_'), 'mov', c(r'v_mov'),
          'bperm', c(r'ds_bpermute'), 'f64', c(r'v_\w+_f64'), 'branch', c(r's_cbranch'))
