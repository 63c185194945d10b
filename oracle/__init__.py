"""CPU oracle for the PLONK prover hot path.  TEST INFRASTRUCTURE ONLY.

This package is an independent big-integer restatement of the reference's
`backend_new` prover (radix-2 NTT, coset NTT, round-3 quotient, KZG commit by
MSM, SHA-256 Fiat-Shamir transcript, 5-round prover, verifier).  Every
function cites the reference file:line it follows.

It is a checker, never the product: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  Nothing under
`plonk_amd/` imports it, and the product fails loudly when its HIP library
is missing.

Parity status: the reference ships no byte-level golden vectors and cannot be
compiled in this environment (Rust, no toolchain).  The oracle is pinned on
every known-answer test the reference's own unit tests hold for this path
(field identities, NTT-vs-Horner, MSM([G,G,G],[1,2,3]) = 6G, f(5) = 86 opening,
prove/verify accept and reject), plus the published BLS12-381 constants
(generator, its ZCash compressed encoding, two-adic root of unity).  The byte
encodings of compressed G1 points and of transcript challenges are "parity
unpinned" against the reference itself: no reference output holds them.
"""
