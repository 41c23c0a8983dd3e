# dev helper: VGPR / scratch / static-LDS use of every kernel in sd_kernels.hip (compiles to assembly in /tmp)
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix='sdres')
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fno-slp-vectorize', '--offload-arch=gfx950', '-I' + root + '/include',
                '--cuda-device-only', '-S', root + '/syconn_amd/csrc/sd_kernels.hip', '-o', tmp + '/k.s'] + sys.argv[1:], check=True,
               stderr=subprocess.DEVNULL)
name = None
rows = {}
for line in open(tmp + '/k.s'):
    m = re.match(r'\s+\.amdhsa_kernel (\S+)', line)
    if m: name = m.group(1); rows[name] = {}
    for k in ('next_free_vgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed_size'):
        m = re.match(r'\s+\.amdhsa_%s (\d+)' % k, line)
        if m and name: rows[name][k] = int(m.group(1))
dem = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.split('\n')
for n, d in zip(rows, dem):
    r = rows[n]
    d = re.sub(r'\(.*', '', d).replace('void ', '')
    print(f"{d:60s} vgpr {r.get('next_free_vgpr', 0):4d} scratch {r.get('private_segment_fixed_size', 0):4d} lds {r.get('group_segment_fixed_size', 0):6d}")
