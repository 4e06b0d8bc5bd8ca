"""The coefficient tables of redner_amd/csrc/libm_exact.h are glibc's: this reads them out of the system's libm.so.6
(glibc 2.35) and checks the header against it, entry by entry.

  python tools/extract_libm_tables.py            # check: every table of the header == the bytes inside libm.so.6
  python tools/extract_libm_tables.py --dump DIR # write the tables as C initialiser lists (hex floats), one file per table

A table is located by the bit pattern of its first entries (no addresses: another build of the same glibc lays its
.rodata out differently), so the check also says whether THIS machine's C library still carries the tables the header
was made from.  2 / pi (`two_over_pi_table`) is additionally recomputed from scratch (Machin's formula, 600 digits).
The routines themselves are restated by hand from the published sources (s_sin.c, branred.c, e_atan2.c, s_atan.c,
e_asin.c, e_log.c, e_pow.c); where the x86-64 `_fma` build fuses a multiply-add was read off `objdump -d` of the same
file (the ifunc'ed variants selected on machines with FMA + AVX2)."""
import os
import re
import struct
import sys
from decimal import Decimal as D, getcontext

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
HEADER = os.path.join(ROOT, 'redner_amd', 'csrc', 'libm_exact.h')
LIBM = os.environ.get('LIBM_SO', '/lib/x86_64-linux-gnu/libm.so.6')
# function in the header -> (entries, entries per row when dumped, kind)
TABLES = {'sincos_table': (440, 4, 'd'), 'atan_table': (241 * 7, 7, 'd'), 'asin_table': (2568, 8, 'd'), 'inroot_table': (128, 8, 'd'),
          'log_table': (256, 4, 'd'), 'pow_log_table': (512, 4, 'd'), 'exp_table': (256, 4, 'Q'), 'two_over_pi_table': (75, 10, 'd')}


def header_tables():
    src = open(HEADER).read()
    out = {}
    for name, (n, _per, kind) in TABLES.items():
        m = re.search(r'%s\(\) \{\s*static const \w+ tab\[[^\]]*\] = \{(.*?)\};' % name, src, re.S)
        assert m, name
        toks = [t.strip() for t in m.group(1).replace('\n', ' ').split(',') if t.strip()]
        if kind == 'Q':
            vals = [int(t.rstrip('ul'), 16) for t in toks]
        else:
            vals = [float.fromhex(t) if 'x' in t else float(t) for t in toks]
        assert len(vals) == n, (name, len(vals), n)
        out[name] = vals
    return out


def pack(vals, kind):
    return struct.pack('<%d%s' % (len(vals), kind), *vals)


def two_over_pi_digits(n):
    getcontext().prec = 640

    def arctan_inv(k):
        x = D(1) / k
        s = t = x
        j = 1
        while True:
            t = -t / (k * k)
            j += 2
            term = t / j
            if abs(term) < D(10) ** -630:
                return s
            s += term
    v = D(2) / (16 * arctan_inv(5) - 4 * arctan_inv(239))
    out = []
    for _ in range(n):
        v *= 1 << 24
        d = int(v)
        v -= d
        out.append(float(d))
    return out


def main():
    blob = open(LIBM, 'rb').read()
    tabs = header_tables()
    dump = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == '--dump' else None
    bad = 0
    for name, (n, per, kind) in TABLES.items():
        want = pack(tabs[name], kind)
        at = blob.find(want[:8 * 6])
        if at < 0:
            print('%-18s NOT FOUND in %s (another glibc?)' % (name, LIBM))
            bad += 1
            continue
        same = blob[at:at + len(want)] == want
        print('%-18s %5d entries at 0x%x of %s: %s' % (name, n, at, os.path.basename(LIBM), 'identical' if same else 'DIFFERS'))
        bad += not same
        if dump:
            vals = struct.unpack('<%d%s' % (n, kind), blob[at:at + len(want)])
            fmt = (lambda v: '0x%016xull' % v) if kind == 'Q' else (lambda v: v.hex())
            os.makedirs(dump, exist_ok=True)
            with open(os.path.join(dump, name + '.inc'), 'w') as f:
                for i in range(0, n, per):
                    f.write('        ' + ', '.join(fmt(v) for v in vals[i:i + per]) + ',\n')
    ok = tabs['two_over_pi_table'] == two_over_pi_digits(75)
    print('two_over_pi_table  == the first 1800 bits of 2 / pi computed here: %s' % ok)
    bad += not ok
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
