"""Instruction mix of the hot loop of a kernel in a --save-temps .s file (gfx950).

    python tools/isa_mix.py <file.s> <kernel name substring> [--top N]

The hot loop is taken to be the LONGEST backward-branch span of the kernel (label ... s_cbranch_* label).  Prints the
count per mnemonic class and the most frequent mnemonics: the marching kernels are bound by the number of instructions
ONE wavefront issues (DESIGN section 4 "Round 4"), so this count is their arithmetic time."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 40
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'^(_Z\w*%s\w*):' % pat, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = lines[start:end + 1]
    labels = {}
    ins = []
    for l in body:
        s = l.split(';')[0].strip()
        if not s or s.startswith('.') and not s.endswith(':'):
            continue
        if s.endswith(':'):
            labels[s[:-1]] = len(ins)
            continue
        ins.append(s)
    best = (0, 0, 0)
    for i, s in enumerate(ins):
        m = re.match(r's_cbranch_\w+\s+(\S+)', s) or re.match(r's_branch\s+(\S+)', s)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    n, a, b = best
    loop = ins[a:b + 1]
    print('kernel %s: %d instructions, hot loop %d' % (lines[start][:-1][:70], len(ins), n))

    def cls(m):
        if m.startswith('v_pk_fma') or m.startswith('v_pk_mul') or m.startswith('v_pk_add'):
            return 'v_pk_' + m.split('_')[2]
        if m.startswith('v_fma') or m.startswith('v_mul_f32') or m.startswith('v_add_f32') or m.startswith('v_sub_f32') or m.startswith('v_mac') or m.startswith('v_fmac'):
            return 'valu f32 scalar-ish (fma/mul/add)'
        if 'dpp' in m:
            return 'dpp'
        if m.startswith('v_mov') or m.startswith('v_accvgpr'):
            return 'v_mov / accvgpr'
        if m.startswith('v_cndmask'):
            return 'v_cndmask'
        if m.startswith('v_readlane') or m.startswith('v_writelane') or m.startswith('v_readfirstlane'):
            return 'lane <-> sgpr'
        if m.startswith('v_'):
            return 'valu other'
        if m.startswith('ds_'):
            return 'lds'
        if m.startswith('buffer_') or m.startswith('global_') or m.startswith('scratch_') or m.startswith('flat_'):
            return 'vmem ' + ('load' if 'load' in m else 'store')
        if m.startswith('s_waitcnt'):
            return 's_waitcnt'
        if m.startswith('s_nop'):
            return 's_nop'
        if m.startswith('s_load') or m.startswith('s_buffer_load'):
            return 'smem'
        if m.startswith('s_'):
            return 'salu'
        return 'other'
    by = collections.Counter()
    mn = collections.Counter()
    dpp = 0
    for s in loop:
        m = s.split()[0]
        c = cls(m)
        if 'dpp' in s and not c == 'dpp':
            c = 'dpp'
        by[c] += 1
        mn[m] += 1
    for c, k in by.most_common():
        print('  %5d  %s' % (k, c))
    print('  top mnemonics:', ', '.join('%s %d' % kv for kv in mn.most_common(top)))


if __name__ == '__main__':
    main()
