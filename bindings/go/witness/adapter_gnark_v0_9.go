// The adapter between Replay.Bind and gnark's compiled constraint system (ADVICE r4: no gnark ConstraintSystem implements HintSystem).
//
// UNCOMPILED here like the rest of bindings/go, and -- unlike the rest, which only needs this repository's own header -- written against
// the constraint package of gnark v0.9.1 (go.mod:5 of the reference) WITHOUT the module at hand (not vendored, no network): the field and
// method names below (System.Instructions / Blueprints / Levels, PackedInstruction.BlueprintID, PackedInstruction.Unpack,
// BlueprintHint.DecompressHint, HintMapping.HintID) are that version's as the author remembers them. The blueprint id lives on the
// PACKED instruction (System.Instructions[i]); the unpacked constraint.Instruction carries only ConstraintOffset / WireOffset / Calldata,
// which is why the interface is indexed by instruction number (ADVICE r5). A maintainer with the module checks
// them with `go vet ./bindings/go/...`; if a name differs, this file is the only place to touch. Nothing here can produce a wrong witness:
// Bind only computes an ORDER, every served record is cross-checked against the call's inputs (replay.go: fits), and Bind compares the
// number of hint calls it finds with the trace layout and fails loudly on any difference.
package witness

import (
	"github.com/consensys/gnark/constraint"
	"github.com/consensys/gnark/constraint/solver"
)

type systemAdapter struct{ sys *constraint.System }

// FromSystem wraps the System embedded in a compiled R1CS / SparseR1CS: witness.FromSystem(&ccs.(*cs_bn254.R1CS).System).
func FromSystem(sys *constraint.System) HintSystem { return systemAdapter{sys} }

func (a systemAdapter) GetNbInstructions() int { return len(a.sys.Instructions) }

func (a systemAdapter) GetLevels() [][]int { return a.sys.Levels }

// GetHintIDAt: instruction i is a hint call when its blueprint is the generic hint blueprint; its calldata decompresses to a HintMapping.
func (a systemAdapter) GetHintIDAt(i int) (solver.HintID, bool) {
	pi := a.sys.Instructions[i]
	bp, ok := a.sys.Blueprints[pi.BlueprintID].(constraint.BlueprintHint)
	if !ok {
		return 0, false
	}
	var hm constraint.HintMapping
	bp.DecompressHint(&hm, pi.Unpack(a.sys))
	return hm.HintID, true
}
