"""Host-side mirror of the reference's `entropy` package (src/entropy/): same module and class names,
arithmetic in HIP kernels through the C ABI."""
