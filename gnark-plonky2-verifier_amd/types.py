"""Mirror of the reference's `types` package: raw (JSON-level) circuit and proof data.

    types.ReadCommonCircuitData          types/common_data.go:61-127
    types.ReadProofWithPublicInputs      types/deserialize.go:92-108
    types.ReadVerifierOnlyCircuitData    types/deserialize.go:110-126

The raw values keep the JSON text; parsing and shape checks happen in libgpv's C++ ingest (csrc/gpv_ingest.cpp), so
the Python layer never touches field elements.
"""
from pathlib import Path


class _Raw:
    def __init__(self, text):
        self.text = text if isinstance(text, bytes) else text.encode()


class CommonCircuitData(_Raw):
    """types.CommonCircuitData (types/types.go:74-86), still in its JSON form."""


class ProofWithPublicInputsRaw(_Raw):
    """types.ProofWithPublicInputsRaw (types/deserialize.go:9-43)"""


class VerifierOnlyCircuitDataRaw(_Raw):
    """types.VerifierOnlyCircuitDataRaw (types/deserialize.go:86-89)"""


def ReadCommonCircuitData(path):
    return CommonCircuitData(Path(path).read_bytes())


def ReadProofWithPublicInputs(path):
    return ProofWithPublicInputsRaw(Path(path).read_bytes())


def ReadVerifierOnlyCircuitData(path):
    return VerifierOnlyCircuitDataRaw(Path(path).read_bytes())
