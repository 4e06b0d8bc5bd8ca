"""Print registers / scratch / LDS / occupancy of every stage kernel (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [file ...]   (default: render.cpp and hip/trace.hip)"""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
CSRC = os.path.join(ROOT, 'redner_amd', 'csrc')
files = sys.argv[1:] or [os.path.join(CSRC, 'render.cpp'), os.path.join(CSRC, 'hip', 'trace.hip')]
for f in files:
    cmd = ['/opt/rocm/bin/hipcc', '-x', 'hip', '--offload-arch=gfx950', '-std=c++17', '-O3', '-fPIC', '-I' + os.path.join(CSRC, 'hip'),
           '-I' + CSRC, '-Wno-unused-result', '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage', '-c', f, '-o', '/dev/null']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    name, rows = None, {}
    for line in err.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1); rows[name] = {}
        for key, short in (('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), (r'ScratchSize \[bytes/lane\]', 'scratch'),
                           (r'Occupancy \[waves/SIMD\]', 'occ'), (r'LDS Size \[bytes/block\]', 'lds'), ('SGPRs', 'sgpr')):
            m = re.search(key + r': (\d+)', line)
            if m and name:
                rows[name][short] = int(m.group(1))
    print(os.path.basename(f))
    for n, v in rows.items():
        short = re.sub(r'^_ZN4exec\d+stage_kernelIN3rdr\d+', '', n)
        short = re.sub(r'EEEvT_i$', '', short)[:44]
        print('  %-44s vgpr %3d agpr %3d scratch %4d lds %6d occ %d' % (short, v.get('vgpr', 0), v.get('agpr', 0), v.get('scratch', 0), v.get('lds', 0), v.get('occ', 0)))
