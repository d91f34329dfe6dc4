"""Host-side mirror of the reference's `cubercnn` package for the Cube R-CNN forward/loss path
(same module / class / registry names, same config keys, same state-dict keys), calling the
MI355X kernels of libcr3dod.so through 3dod_amd.hipops.  Put `3dod_amd/` on PYTHONPATH to make
`import cubercnn` resolve here."""
