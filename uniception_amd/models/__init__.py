"""Host-side mirror of the reference's ``uniception.models`` interface for the DUSt3R two-view path:
same class names, constructor signatures, dataclasses and state_dict keys; the compute is the HIP
kernel library (uniception_amd.ops / uniception_amd.engine)."""
