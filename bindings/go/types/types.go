// Package types mirrors the reference's raw (JSON-level) records (types/types.go, types/deserialize.go, types/common_data.go).
// UNCOMPILED here (no Go toolchain in the build image). The raw forms stay JSON text: libgpv's arena parser
// (csrc/gpv_ingest.cpp) is the deserialiser, so nothing is parsed twice.
package types

import "os"

type CommonCircuitData struct{ JSON []byte }          // types/types.go:62-86
type ProofWithPublicInputsRaw struct{ JSON []byte }   // types/deserialize.go:40-43
type VerifierOnlyCircuitDataRaw struct{ JSON []byte } // types/deserialize.go:110-113

func read(path string) []byte {
	b, err := os.ReadFile(path)
	if err != nil {
		panic(err) // the reference panics on unreadable files too (types/deserialize.go:93-96)
	}
	return b
}

func ReadCommonCircuitData(path string) CommonCircuitData { return CommonCircuitData{read(path)} } // types/common_data.go:61
func ReadProofWithPublicInputs(path string) ProofWithPublicInputsRaw { // types/deserialize.go:92
	return ProofWithPublicInputsRaw{read(path)}
}
func ReadVerifierOnlyCircuitData(path string) VerifierOnlyCircuitDataRaw { // types/deserialize.go:110
	return VerifierOnlyCircuitDataRaw{read(path)}
}
