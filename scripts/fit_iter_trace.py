"""kernel sequence of single fit iterations from a rocprofv3 trace of `bench.py --mode fit`:
python scripts/fit_iter_trace.py <results.db> [iteration numbers...]   (an iteration = the launches between two fit_adam_kernel)"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
want = [int(x) for x in sys.argv[2:]] or [50, 130, 250]
def short(n):
    n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "")).replace("at::native::", "")
    n = re.sub(r"vectorized_elementwise_kernel<\d+, ", "vec<", n)
    n = re.sub(r"elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl_nocast<", "ew_nocast<", n)
    n = re.sub(r"std::array<char\*, \d+ul>", "", n)
    return n[:150]
it, cur = 0, []
for name, s, e in rows:
    cur.append((short(name), s, e))
    if "fit_adam_kernel" in name:
        if it in want:
            t0 = cur[0][1]
            print(f"==== iteration {it}: {len(cur)} launches, {(cur[-1][2] - t0) / 1e3:.0f} us from first start to last end, kernel time {sum(e - s for _, s, e in cur) / 1e3:.0f} us")
            for n, s, e in cur:
                print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}  {n}")
        it += 1; cur = []
print("iterations seen:", it)
