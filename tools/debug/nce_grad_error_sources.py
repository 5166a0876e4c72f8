"""CPU, f64: which rounding owns the error of the InfoNCE feature gradient (simclr.py: _InfoNCE.backward)?
Rows as a trunk produces them (a common direction + noise: cosines of 0.92 ... 0.998) and uncorrelated rows.  Findings (round 4):
the bf16 rounding of the normalised rows fn as an operand of  d fn = (2 / T) P fn  owns it (3.4e-2 at cos 0.998), because the
normalisation backward keeps only the part of d fn orthogonal to fn; rounding P = d loss / d sim contributes ~1e-3, rounding fn in
the similarity product ~1e-3; fn as hi + lo bf16 terms in the gradient product brings the total to ~1e-3."""
import torch


def r(t):
    return t.to(torch.bfloat16).float()


def main():
    torch.manual_seed(1)
    for n, p, eps in ((128, 768, None), (128, 768, 0.3), (128, 768, 0.05), (1024, 2048, 0.1)):
        if eps is None:
            f = torch.randn(n, p, dtype=torch.double)
        else:
            f = torch.randn(1, p, dtype=torch.double) + eps * torch.randn(n, p, dtype=torch.double)
        T = 0.1
        inv = 1.0 / f.norm(dim=1, keepdim=True)
        fn = f * inv

        def grad(fn_round, p_round, fn2_round, p_hilo=False, fn_hilo=False):
            fa = r(fn.float()).double() if fn_round else fn
            sim = fa @ fa.t() / T
            idx = torch.arange(n)
            eye = torch.eye(n, dtype=torch.bool)
            pos = (idx[:, None] - idx[None, :]).abs() == 1
            neg = ~eye & ~pos
            lse = torch.logsumexp(sim[neg], 0)
            W = torch.zeros(n, n, dtype=torch.double)
            W[neg] = torch.exp(sim[neg] - lse)
            W[pos] = -1.0 / pos.sum()
            if p_round:
                Wh = r(W.float()).double()
                W = Wh + (r((W - Wh).float()).double() if p_hilo else 0)
            fb = fn
            if fn2_round:
                fb = r(fn.float()).double()
                if fn_hilo:
                    fb = fb + r((fn - fb).float()).double()
            dfn = (2.0 / T) * (W @ fb)
            return inv * (dfn - fn * (fn * dfn).sum(1, keepdim=True))

        ref = grad(False, False, False)
        print(f"{n} x {p}, rows {'uncorrelated' if eps is None else 'common + %.2f noise' % eps}: mean cosine {float((fn @ fn.t()).mean()):.4f}")
        for name, args in (("fn bf16 in the similarity product only", (True, False, False)), ("P bf16 only", (False, True, False)),
                           ("fn bf16 in the gradient product only", (False, False, True)), ("all three (round 3's kernel)", (True, True, True)),
                           ("+ P as hi + lo", (True, True, True, True)), ("+ fn as hi + lo in the gradient product (round 4)", (True, True, True, False, True))):
            g = grad(*args)
            print(f"    {name:52s} {float((g - ref).norm() / ref.norm()):.2e}")


if __name__ == "__main__":
    main()
