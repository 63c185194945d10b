"""plonk_amd -- MI355X-native PLONK prover hot path (NTT, quotient, KZG MSM) behind the
interface of the reference's `backend_new` crate.  All arithmetic runs in libplonk_hip.so
(hand-written HIP for gfx950); this package only marshals buffers.  There is no CPU path.
"""
from .api import (Context, Domain, Srs, Commitment, ProvingKey, Prover, Proof, fft, ifft, coset_fft, coset_ifft,
                  msm, fr_to_array, fr_from_array, PlonkError, VerificationKey, Verifier, g2_generator, g2_mul,
                  pairing_check, Polynomial, OpeningProof, BatchOpening)

__all__ = ["Context", "Domain", "Srs", "Commitment", "ProvingKey", "Prover", "Proof", "fft", "ifft", "coset_fft",
           "coset_ifft", "msm", "fr_to_array", "fr_from_array", "PlonkError", "VerificationKey", "Verifier", "g2_generator",
           "g2_mul", "pairing_check", "Polynomial", "OpeningProof", "BatchOpening"]
