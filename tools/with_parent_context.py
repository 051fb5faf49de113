import subprocess, sys, torch
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
sys.exit(subprocess.run(sys.argv[1:]).returncode)
