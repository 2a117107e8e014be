"""RNN-Transducer pieces of the reference (espnet/nets/pytorch_backend/transducer/*) on the HIP kernels."""
