// Package variables mirrors the reference's circuit-level records (variables/circuit.go, variables/deserialize.go).
// UNCOMPILED here (no Go toolchain). In the reference these are trees of frontend.Variable; here a proof is a packed record
// (wire format of include/gpv.h) and a batch is n of them back to back.
package variables

import (
	"sync"

	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/types"
)

// VerifierOnlyCircuitData (variables/circuit.go:21-24): constants/sigmas cap + circuit digest, kept as the JSON the circuit
// handle is built from.
type VerifierOnlyCircuitData struct{ Raw types.VerifierOnlyCircuitDataRaw }

func DeserializeVerifierOnlyCircuitData(raw types.VerifierOnlyCircuitDataRaw) VerifierOnlyCircuitData { // deserialize.go:149
	return VerifierOnlyCircuitData{raw}
}

// Proof (variables/circuit.go:8-14) and PublicInputs travel together in the packed record: ProofWithPublicInputs.Proof is the
// whole record, PublicInputs a view of its public-input words. A value holds N >= 1 proofs of one circuit.
type Proof struct {
	Circuit *gpv.Circuit
	Packed  []byte
	N       int
}
type ProofWithPublicInputs struct { // variables/circuit.go:16-19
	Proof        Proof
	PublicInputs []uint64 // [N][num_public_inputs], as stored in the record (not reduced: verifier.go:84-141 exempts them)
}

var (
	circuitsMu sync.Mutex // goroutines deserialise concurrently; a gpv_circuit itself is immutable and shareable (include/gpv.h, Threading)
	circuits   = map[string]*gpv.Circuit{}
)

// CircuitFor returns the (cached) gpv_circuit for a pair of circuit data. Safe for concurrent use.
func CircuitFor(common types.CommonCircuitData, vo VerifierOnlyCircuitData) *gpv.Circuit {
	key := string(common.JSON) + "\x00" + string(vo.Raw.JSON)
	circuitsMu.Lock()
	defer circuitsMu.Unlock()
	if c, ok := circuits[key]; ok {
		return c
	}
	c := gpv.NewCircuit(common.JSON, vo.Raw.JSON)
	circuits[key] = c
	return c
}

func publicInputs(c *gpv.Circuit, packed []byte, n int) []uint64 {
	d := c.Dims()
	numPI, offPI := d.NumPublicInputs, d.OffPublicInputs
	rec := c.ProofNBytes() / 8
	out := make([]uint64, 0, n*numPI)
	for i := 0; i < n; i++ {
		for k := 0; k < numPI; k++ {
			w := 8 * (i*rec + offPI + k)
			var v uint64
			for b := 0; b < 8; b++ {
				v |= uint64(packed[w+b]) << (8 * b)
			}
			out = append(out, v)
		}
	}
	return out
}

// DeserializeProofWithPublicInputs (variables/deserialize.go:114-147): shape errors panic like the reference (fri_utils.go:167-228).
func DeserializeProofWithPublicInputs(raw types.ProofWithPublicInputsRaw, c *gpv.Circuit) ProofWithPublicInputs {
	packed := c.PackProof(raw.JSON)
	return ProofWithPublicInputs{Proof{c, packed, 1}, publicInputs(c, packed, 1)}
}

// DeserializeProofsWithPublicInputs packs n JSON proofs on nThreads host threads (gpv_proof_pack_json_batch).
func DeserializeProofsWithPublicInputs(raws []types.ProofWithPublicInputsRaw, c *gpv.Circuit, nThreads int) ProofWithPublicInputs {
	js := make([][]byte, len(raws))
	for i := range raws {
		js[i] = raws[i].JSON
	}
	packed := c.PackProofs(js, nThreads)
	return ProofWithPublicInputs{Proof{c, packed, len(raws)}, publicInputs(c, packed, len(raws))}
}
