"""Stated tolerances of the parity tests (one place, so that a bound is never looser in one file than in another).

fp32 mode: BASELINE.json north_star -- decoder logits within 1e-4 max-abs of the reference arithmetic (the fp64 oracle); measured
1.3e-7 at C2 size.  Because random-init logits are ~0.03 (max ~0.08-0.15) the absolute bound alone is loose, so fp32 logits are also
held to 1e-4 of the LARGEST reference logit.
bf16 mode (bf16 operands, fp32 accumulate): measured 3.8e-4 .. 6.2e-4 max-abs against the fp64 oracle over C3 / C4 / He = 512 / small
shapes (round 3 logs); the bound is ~3x that, absolute AND relative to the largest reference logit (measured 0.4-0.8 %) -- a decoder that
returned zeros, or logits of the wrong scale, fails both."""
F32_LOGIT_TOL = 1e-4
F32_LOGIT_REL = 1e-4
BF16_LOGIT_TOL = 2e-3
BF16_LOGIT_REL = 2.5e-2


def check_logits(got, ref, compute, what=""):
    """got, ref: torch tensors of the same shape (ref float64).  Returns (max-abs error, largest |reference logit|)."""
    e = (got.double() - ref.double()).abs().max().item()
    top = ref.double().abs().max().item()
    tol, rel = (F32_LOGIT_TOL, F32_LOGIT_REL) if compute == "f32" else (BF16_LOGIT_TOL, BF16_LOGIT_REL)
    assert e < tol, (what, compute, "max-abs", e, tol)
    assert e < rel * top, (what, compute, "relative to the largest reference logit", e, top, rel)
    return e, top
