// The reference's verifier/verifier_test.go:13-41 against the shim -- same files, same call sequence; the only differences are the
// context in place of the gnark test engine and the accept bits in place of test.IsSolved. UNCOMPILED here (no Go toolchain).
package verifier

import (
	"testing"

	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/types"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/variables"
)

// TestStepVerifier is the reference's test of that name on the fixture it uses (verifier/verifier_test.go:13-41: "step", :17);
// TestBlockVerifier runs the same sequence on the other fixture under testdata/ (the one fri_test.go / plonk_test.go also use).
func TestStepVerifier(t *testing.T)  { runVerifier(t, "step") }
func TestBlockVerifier(t *testing.T) { runVerifier(t, "decode_block") }

func runVerifier(t *testing.T, plonky2Circuit string) {
	ctx := gpv.NewContext(0)
	defer ctx.Close()

	commonCircuitData := types.ReadCommonCircuitData("../../../tests/golden/" + plonky2Circuit + "/common_circuit_data.json")
	verifierOnlyCircuitData := variables.DeserializeVerifierOnlyCircuitData(types.ReadVerifierOnlyCircuitData("../../../tests/golden/" + plonky2Circuit + "/verifier_only_circuit_data.json"))
	circuit := variables.CircuitFor(commonCircuitData, verifierOnlyCircuitData)
	proofWithPis := variables.DeserializeProofWithPublicInputs(types.ReadProofWithPublicInputs("../../../tests/golden/"+plonky2Circuit+"/proof_with_public_inputs.json"), circuit)

	verifierChip := NewVerifierChip(ctx, commonCircuitData)
	accept := verifierChip.Verify(proofWithPis.Proof, proofWithPis.PublicInputs, verifierOnlyCircuitData)
	if len(accept) != 1 || !accept[0] {
		t.Fatal("the fixture proof was rejected")
	}

	// one flipped bit in an opening: rejected, with an error-free call (a rejected proof is not an error)
	bad := proofWithPis.Proof
	bad.Packed = append([]byte(nil), bad.Packed...)
	bad.Packed[8*3] ^= 1
	if verifierChip.Verify(bad, nil, verifierOnlyCircuitData)[0] {
		t.Fatal("a corrupted proof was accepted")
	}
}
