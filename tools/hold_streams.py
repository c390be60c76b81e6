"""An idle process that holds N HIP streams for T seconds (tools/hold_streams.py N T): run beside the drop-in to see whether another process's hardware queues slow it
(round 5: they do not -- 1.23 s against 1.22 s for the 12.5 M-read leg with 16 streams held)."""
import torch, time, sys
torch.cuda.init(); x = torch.zeros(1, device="cuda")
ss = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]
for s in ss:
    with torch.cuda.stream(s): y = x + 1
torch.cuda.synchronize()
print("holding", len(ss), "streams", flush=True)
time.sleep(float(sys.argv[2]))
