"""run a pytest selection with the caching allocator's free blocks full of garbage (a read of memory nobody wrote then sees different
bytes in different buffers instead of the zeros of fresh pages): python scripts/poison_then_test.py <GiB> <pytest args ...>"""
import sys, torch, pytest
gib = int(sys.argv[1])
blocks = [torch.empty(1 << 28, dtype=torch.float32, device="cuda") for _ in range(gib)]      # 1 GiB each
for i, b in enumerate(blocks):
    b.uniform_(-3.0, 3.0)
    b[::7] = float("nan") if i % 2 else 1e30
del blocks
small = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22) for _ in range(16)]
del small
torch.cuda.synchronize()
sys.exit(pytest.main(sys.argv[2:]))
