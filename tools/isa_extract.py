"""Developer helper (no GPU): compile one csrc/*.hip to gfx950 assembly with build.py's flags and print, for the kernel whose mangled
name contains <substr>, the instruction mix of the whole kernel and of every loop (by the assembler's loop annotations).
usage: tools/isa_extract.py render_fused.hip 'render_fwd_kernelILi10ELi8ELi8ELi2ELb1ELb1' [-D...] [--dump out.s]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-munsafe-fp-atomics', '-fno-gpu-flush-denormals-to-zero']


def kinds(lines):
    c = dict(valu=0, pk=0, salu=0, smem=0, vmem=0, lds=0, wait=0, branch=0)
    for x in lines:
        x = x.strip()
        if x.startswith('v_'):
            c['valu'] += 1
            if x.startswith('v_pk_'): c['pk'] += 1
        elif x.startswith('s_load') or x.startswith('s_buffer_load'): c['smem'] += 1
        elif x.startswith('s_waitcnt'): c['wait'] += 1
        elif x.startswith('s_cbranch') or x.startswith('s_branch'): c['branch'] += 1
        elif x.startswith('s_'): c['salu'] += 1
        elif x.startswith('ds_'): c['lds'] += 1
        elif x.startswith('global_') or x.startswith('scratch_') or x.startswith('buffer_') or x.startswith('flat_'): c['vmem'] += 1
    return c


def main():
    src, sub = sys.argv[1], sys.argv[2]
    defs = [a for a in sys.argv[3:] if a.startswith('-D')]
    dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
    out = '/tmp/_isa_%s.s' % os.path.basename(src)
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + defs + ['-S', '--cuda-device-only', '-o', out, os.path.join(ROOT, 'differentiable-blocksworld_amd', 'csrc', src)],
                          stderr=subprocess.DEVNULL)
    L = open(out).read().split('\n')
    start = next(i for i, l in enumerate(L) if l.endswith(':') is False and re.match(r'^_Z\S*' + re.escape(sub) + r'\S*:', l))
    end = next(i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end'))
    K = L[start:end]
    if dump: open(dump, 'w').write('\n'.join(K))
    print('kernel:', K[0][:120])
    print('whole kernel:', kinds(K))
    for l in L[end:end + 120]:
        if re.search(r'; (NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize)', l): print('  ', l.strip())
    # loops: blocks annotated "in Loop: Header=BBx_y Depth=d" / "Loop Header"
    cur, blocks = None, {}
    for l in K:
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l)
        if m:
            ann = m.group(2) or ''
            h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', ann)
            if 'Loop Header' in ann:
                d = re.search(r'Depth=(\d+)', ann).group(1)
                cur = ('.L' + m.group(1)[2:], d) if False else (m.group(1)[2:], d)
            elif h: cur = (h.group(1), h.group(2))
            else: cur = None
            continue
        if cur: blocks.setdefault(cur, []).append(l)
    for (h, d), ls in blocks.items():
        print('loop %-10s depth %s:' % (h, d), kinds(ls))


main()
