"""Host-side mirror of the reference's lib/layer_utils for the forward path: same module
names, function names, argument meaning and return types, backed by the HIP library."""
