"""Host-side mirrors of the reference's `models/` interfaces for the hot path (same names, argument meaning and error
behaviour); all arithmetic is delegated to the HIP engine behind the C ABI."""
