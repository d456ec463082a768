import sys, os, tempfile, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_world2 as T

if __name__ == "__main__":
    tmp = pathlib.Path(tempfile.mkdtemp())
    r0, r1 = T._two_ranks(tmp, T._train, sys.argv[1] if len(sys.argv) > 1 else "replicate")
    arg = sys.argv[1] if len(sys.argv) > 1 else "replicate"
    single, losses = T._train(lambda s: slice(s * 2 * T.B, (s + 1) * 2 * T.B), world=1,
                              tables="replicate+l2" if arg.endswith("+l2") else "replicate")
    sparse, _, _ = T._data()
    print("losses r0", r0["losses"], "\nlosses r1", r1["losses"], "\nsingle", losses)
    for k, want in single.items():
        a, b, w = r0["sd"][k].numpy(), r1["sd"][k].numpy(), want.numpy()
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(w)
        bad2 = np.abs(a - w) > 3e-4 + 1e-3 * np.abs(w)
        print(f"{k:40s} shape {a.shape} bad(a,b)={bad.sum():5d} max|a-b|={np.abs(a-b).max():.3e}  bad(a,single)={bad2.sum():5d} max|a-w|={np.abs(a-w).max():.3e}")
        if bad.sum() and a.ndim == 2 and "embed_dict" in k:
            rows = np.unique(np.nonzero(bad)[0])
            f = int(k.split(".C")[1].split(".")[0])
            cnt = np.bincount(sparse[:, f].numpy(), minlength=a.shape[0])
            print("   bad rows", rows[:20], "lookups/row over all steps", cnt[rows][:20])
            r = rows[0]
            print("   row", r, "a", a[r][:6], "\n          b", b[r][:6], "\n          w", w[r][:6])
