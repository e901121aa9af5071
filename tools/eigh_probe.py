import torch, time
torch.manual_seed(0)
for n in (64, 93, 120, 160, 224, 311, 371, 700, 1300):
    A = torch.randn(n, n, dtype=torch.float64, device="cuda"); A = A + A.T
    for _ in range(2): torch.linalg.eigh(A)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): w, V = torch.linalg.eigh(A)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 5
    print(f"torch.linalg.eigh (rocSOLVER syevd) n={n}: {el*1e3:.2f} ms", flush=True)
import numpy as np
for n in (120, 311, 371):
    A = np.random.randn(n, n); A = A + A.T
    t = time.perf_counter()
    for _ in range(5): np.linalg.eigh(A)
    print(f"numpy eigh n={n}: {(time.perf_counter()-t)/5*1e3:.2f} ms")
