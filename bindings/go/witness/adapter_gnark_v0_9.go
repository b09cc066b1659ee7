// The adapter between Replay.Bind and gnark's compiled constraint system (ADVICE r4: no gnark ConstraintSystem implements HintSystem).
//
// UNCOMPILED here like the rest of bindings/go, and -- unlike the rest, which only needs this repository's own header -- written against
// the constraint package of gnark v0.9.1 (go.mod:5 of the reference) WITHOUT the module at hand (not vendored, no network): the field and
// method names below (System.Instructions / Blueprints / Levels, PackedInstruction.BlueprintID, System.GetInstruction,
// BlueprintHint.DecompressHint, HintMapping.HintID) are that version's as the author remembers them. A maintainer with the module checks
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

func (a systemAdapter) GetInstruction(i int) constraint.Instruction { return a.sys.GetInstruction(i) }

func (a systemAdapter) GetLevels() [][]int { return a.sys.Levels }

// GetHintIDOf: a hint call is an instruction whose blueprint is the generic hint blueprint; its calldata decompresses to a HintMapping.
func (a systemAdapter) GetHintIDOf(inst constraint.Instruction) (solver.HintID, bool) {
	bp, ok := a.sys.Blueprints[inst.BlueprintID].(constraint.BlueprintHint)
	if !ok {
		return 0, false
	}
	var hm constraint.HintMapping
	bp.DecompressHint(&hm, inst)
	return hm.HintID, true
}
