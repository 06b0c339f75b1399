"""zinc_amd: MI355X-native Groth16 prover backend for zinc-vm (HIP kernels behind a C ABI)."""
from .groth16 import (Context, Parameters, Proof, ProvingAssignment, R1CS, SynthesisError,  # noqa: F401
                      create_proof, create_random_proof)
