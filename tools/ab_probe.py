# dev probe: A/B two builds of the library in ONE gpurun call (boxes differ by a few % in clocks): interleaved runs
import os, subprocess, sys
libs = sys.argv[1:]
for rep in range(2):
    for l in libs:
        out = subprocess.run([sys.executable, 'tools/batch_probe.py'], env=dict(os.environ, SD_LIB_NAME=l, SD_PROBE_BATCHES='1,8'),
                             capture_output=True, text=True).stdout
        print(l, '|', ' | '.join(x.split(': ', 1)[1] for x in out.strip().split('\n') if '1 stream' in x))
