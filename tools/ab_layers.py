"""GPU box: A/B of two (or more) builds of the library in ONE call -- per-op times of a model, interleaved subprocesses (boxes differ
by a few per cent in clocks).  usage: ab_layers.py <arch> <act> <lib name in syconn_amd/> [<lib name> ...]"""
import os
import subprocess
import sys

arch, act, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
for rep in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, 'tools/layer_times_env.py', arch, act, ''], env=dict(os.environ, SD_LIB_NAME=lib),
                             capture_output=True, text=True).stdout
        print(f'{lib:28s}', out.strip().split('\n')[-1][28:])
