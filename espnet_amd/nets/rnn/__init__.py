"""RNN path of the reference (espnet/nets/pytorch_backend/rnn/*) on the espnet_amd HIP kernels."""
