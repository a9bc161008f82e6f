"""quickstep_amd — MI355X (gfx950) execution kernel for Quickstep's vectorized
relational_operators hot path (hash join build/probe, group-by aggregation,
select) behind the C ABI of ``include/qsx.h``.

``quickstep_amd.capi`` is the ctypes binding of that C ABI; it loads the
in-tree ``quickstep_amd/lib/libqsx.so`` and raises if the library is missing —
there is no CPU implementation to fall back to.
"""

__version__ = "0.1.0"
