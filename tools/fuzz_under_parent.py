"""GPU box: the GPU fuzz leg as a child of a process that holds a context on the same GPU (what pytest is for the test)."""
import os, subprocess, sys, torch
x = torch.zeros(1 << 20, device='cuda')      # the parent holds a context on the GPU, like pytest does
torch.cuda.synchronize()
env = dict(os.environ, MALLOC_MMAP_THRESHOLD_='1024', MALLOC_PERTURB_='255', FUZZ_STRIDE='8', FUZZ_OFFSET='5')
env.update(dict(kv.split('=') for kv in sys.argv[1:]))
out = subprocess.run([sys.executable, 'test_fuzz_parity.py', 'gpu'], env=env, capture_output=True, text=True).stdout
print([l[:600] for l in out.splitlines() if l.startswith(('FUZZ', 'ORACLE'))])
