// Package witness replays a hint trace produced on the GPU (gpv.WitnessVerify, include/gpv.h gpv_witness_verify) into gnark's solver:
// the last mile of SURVEY 8f.3. UNCOMPILED here (no Go toolchain, gnark not vendored), like the rest of bindings/go.
//
// What it replaces. The reference registers four hint functions with gnark's solver (goldilocks/base.go:54-59: MulAddHint, ReduceHint,
// InverseHint, SplitLimbsHint; bodies :223-359). While the solver computes the witness of the verifier circuit it calls them 448 677
// times per `step` proof (big.Int multiplications and divisions by p). gpv_witness_verify computes all of those outputs on the GPU, per
// proof, in CALL order of VerifierChip.Verify (verifier/verifier.go:143-178). A Replay hands them to the solver instead of recomputing
// them.
//
// What a hint function is given -- and the constraint that follows. gnark calls a hint with (modulus, inputs, outputs): NO call index.
// And the solver does not run instructions in program order: it solves level by level (constraint.System.Levels: the instructions whose
// inputs are all known), and by default it splits a level over parallel tasks. So "pop the next record" is valid only if
//
//   (1) the solver runs one task (solver.WithNbTasks(1); Options() sets it) -- within a level, instructions then run in ascending index, and
//   (2) the replay knows the order in which THAT solver reaches the hint instructions: Bind walks the compiled system's Levels and maps
//       the j-th EXECUTED hint call to its number in PROGRAM order (hint instructions are emitted in call order: api.Compiler().NewHint
//       appends one instruction per call, base.go:197,262,298,371), i.e. to its record in the trace.
//
// What breaks otherwise: with parallel tasks two hints of one level race for the cursor; without the level map the second level-0 hint
// of the program would be handed the record of the first level-1 hint. Neither can produce a WRONG witness here, because every popped
// record is cross-checked against the hint's own inputs (one multiplication instead of a division) and a record that does not fit is
// discarded in favour of the reference's computation (Mismatches counts them: it must stay 0, anything else means the order assumption
// is violated -- another gnark version, a changed circuit -- and the replay is only costing time). The trace is an accelerator, never an
// authority: the circuit's constraints check every hinted value anyway (base.go:196-213, :246-281, :297-313, :362-400).
package witness

import (
	"fmt"
	"math/big"
	"sync"

	"github.com/consensys/gnark/constraint/solver"
	gl "github.com/succinctlabs/gnark-plonky2-verifier/goldilocks"
)

// hint kinds of include/gpv.h (GPV_HINT_*) and the words each record holds
const (
	kindMulAdd = 0 // (quotient, remainder)
	kindReduce = 1 // (quotient as 4 little-endian words, remainder)
	kindInverse = 2 // (inverse)
	kindSplit = 3 // (hi, lo)
)

var recordWords = [4]int{2, 5, 1, 2}

// Replay serves ONE proof's trace to ONE solver run.
type Replay struct {
	mu         sync.Mutex
	trace      []uint64 // gpv.WitnessVerify: [WitnessVerifyWords] of this proof
	kinds      []uint8  // Circuit.WitnessVerifyLayout: one GPV_HINT_* id per hint call, program order
	offset     []int    // word offset of record k (program order)
	execToProg []int    // j-th executed hint call -> its number in program order (nil until Bind / BindFrom / UseProgramOrder)
	bound      bool     // an order has been chosen: Options() refuses to serve before that
	next       int      // hint calls served so far
	Mismatches int      // popped records that did not fit the call's inputs (recomputed the reference's way)
}

// NewReplay: trace and kinds of one proof (gpv.Context.WitnessVerify / gpv.Circuit.WitnessVerifyLayout).
func NewReplay(trace []uint64, kinds []uint8) *Replay {
	off := make([]int, len(kinds)+1)
	for k, kind := range kinds {
		off[k+1] = off[k] + recordWords[kind]
	}
	if off[len(kinds)] != len(trace) {
		panic(fmt.Sprintf("witness: the layout describes %d words, the trace has %d", off[len(kinds)], len(trace)))
	}
	return &Replay{trace: trace, kinds: kinds, offset: off}
}

// Bind computes the order in which a one-task solver reaches the hint instructions of the compiled verifier circuit: levels in order,
// instructions of a level in ascending index (constraint.System.Levels). Call it once per compiled circuit; the map can be shared by
// every Replay of that circuit (BindFrom). Only the reference's four hints are counted: gnark's own hints (api.ToBinary inside
// BN254Chip.ToVec, the index decompositions) are not in the trace and keep their own functions.
//
// HintSystem is what Bind needs from the compiled circuit. No gnark type implements it as is (ADVICE r4): FromSystem
// (adapter_gnark_v0_9.go) wraps gnark v0.9.1's *constraint.System -- r.Bind(witness.FromSystem(&ccs.(*cs_bn254.R1CS).System)).
type HintSystem interface {
	GetNbInstructions() int
	GetHintIDAt(i int) (solver.HintID, bool) // instruction i's hint id when it is a hint call (the blueprint id is a field of the PACKED instruction)
	GetLevels() [][]int
}

func (r *Replay) Bind(sys HintSystem) error {
	ours := map[solver.HintID]bool{
		solver.GetHintID(gl.MulAddHint): true, solver.GetHintID(gl.ReduceHint): true,
		solver.GetHintID(gl.InverseHint): true, solver.GetHintID(gl.SplitLimbsHint): true,
	}
	if sys == nil {
		return fmt.Errorf("witness: nil system")
	}
	progNumber := make(map[int]int) // instruction index -> hint number in program order
	n := 0
	for i := 0; i < sys.GetNbInstructions(); i++ {
		if id, isHint := sys.GetHintIDAt(i); isHint && ours[id] {
			progNumber[i] = n
			n++
		}
	}
	if n != len(r.kinds) {
		return fmt.Errorf("witness: the circuit calls the reference's hints %d times, the trace layout lists %d", n, len(r.kinds))
	}
	order := make([]int, 0, n)
	for _, level := range sys.GetLevels() {
		for _, i := range level { // ascending within a level
			if k, isOurs := progNumber[int(i)]; isOurs {
				order = append(order, k)
			}
		}
	}
	r.execToProg = order
	r.bound = true
	return nil
}

// BindFrom shares the order map computed by another Replay of the same compiled circuit.
func (r *Replay) BindFrom(other *Replay) { r.execToProg, r.bound = other.execToProg, other.bound }

// UseProgramOrder declares that the solver reaches the hint calls in program order (a solver that does not schedule by levels). With
// gnark's level-scheduled solver this is WRONG for all but the first level: nearly every record would be rejected by the cross-check and
// recomputed the reference's way -- correct, and slower than no replay at all (ADVICE r4). It exists for such solvers and for tests; it is
// never assumed silently.
func (r *Replay) UseProgramOrder() { r.execToProg, r.bound = nil, true }

// Options: the solver options to prove with -- groth16.Prove(ccs, pk, w, backend.WithSolverOptions(replay.Options()...)).
func (r *Replay) Options() []solver.Option {
	if !r.bound {
		// Before round 5 an unbound Replay silently fell back to program order (see UseProgramOrder for what that costs).
		panic("witness: Options() before a successful Bind / BindFrom (or an explicit UseProgramOrder): the replay does not know the solver's hint order")
	}
	return []solver.Option{
		solver.WithNbTasks(1), // the cursor below is an ORDER: see the package comment
		solver.OverrideHint(solver.GetHintID(gl.MulAddHint), r.serve(kindMulAdd, gl.MulAddHint)),
		solver.OverrideHint(solver.GetHintID(gl.ReduceHint), r.serve(kindReduce, gl.ReduceHint)),
		solver.OverrideHint(solver.GetHintID(gl.InverseHint), r.serve(kindInverse, gl.InverseHint)),
		solver.OverrideHint(solver.GetHintID(gl.SplitLimbsHint), r.serve(kindSplit, gl.SplitLimbsHint)),
	}
}

// record of the next executed hint call, or nil when the trace cannot serve it (exhausted / another kind at that position)
func (r *Replay) pop(kind int) []uint64 {
	r.mu.Lock()
	defer r.mu.Unlock()
	j := r.next
	r.next++
	if r.execToProg != nil {
		if j >= len(r.execToProg) {
			return nil
		}
		j = r.execToProg[j]
	}
	if j >= len(r.kinds) || int(r.kinds[j]) != kind {
		return nil
	}
	return r.trace[r.offset[j]:r.offset[j+1]]
}

func (r *Replay) serve(kind int, reference solver.Hint) solver.Hint {
	return func(mod *big.Int, inputs []*big.Int, outputs []*big.Int) error {
		rec := r.pop(kind)
		if rec != nil && fits(kind, rec, inputs) {
			fill(kind, rec, outputs)
			return nil
		}
		r.mu.Lock()
		r.Mismatches++
		r.mu.Unlock()
		return reference(mod, inputs, outputs) // the reference's own computation: always right, only slower
	}
}

var p = gl.MODULUS

func u(x uint64) *big.Int { return new(big.Int).SetUint64(x) }

func words4(w []uint64) *big.Int { // four little-endian 64-bit words
	v := new(big.Int)
	for i := 3; i >= 0; i-- {
		v.Lsh(v, 64).Or(v, u(w[i]))
	}
	return v
}

// fits: does the record answer THIS call? One multiplication by p instead of the hint's division -- and the guard that makes a
// mis-ordered solver harmless.
func fits(kind int, rec []uint64, in []*big.Int) bool {
	switch kind {
	case kindMulAdd: // base.go:223-243: a*b + c = q*p + r, r < p
		if len(in) != 3 || u(rec[1]).Cmp(p) >= 0 {
			return false
		}
		lhs := new(big.Int).Mul(in[0], in[1])
		lhs.Add(lhs, in[2])
		rhs := new(big.Int).Mul(u(rec[0]), p)
		return lhs.Cmp(rhs.Add(rhs, u(rec[1]))) == 0
	case kindReduce: // base.go:284-294: x = q*p + r, r < p
		if len(in) != 1 || u(rec[4]).Cmp(p) >= 0 {
			return false
		}
		rhs := new(big.Int).Mul(words4(rec[:4]), p)
		return in[0].Cmp(rhs.Add(rhs, u(rec[4]))) == 0
	case kindInverse: // base.go:316-336: x * inv = 1 mod p; the inverse of 0 is 0
		if len(in) != 1 || u(rec[0]).Cmp(p) >= 0 {
			return false
		}
		if in[0].Sign() == 0 {
			return rec[0] == 0
		}
		prod := new(big.Int).Mul(in[0], u(rec[0]))
		return prod.Mod(prod, p).Cmp(big.NewInt(1)) == 0
	default: // base.go:339-359: x = hi * 2^32 + lo, both below 2^32
		if len(in) != 1 || rec[0]>>32 != 0 || rec[1]>>32 != 0 {
			return false
		}
		return in[0].Cmp(u(rec[0]<<32|rec[1])) == 0
	}
}

func fill(kind int, rec []uint64, out []*big.Int) {
	switch kind {
	case kindMulAdd:
		out[0].SetUint64(rec[0])
		out[1].SetUint64(rec[1])
	case kindReduce:
		out[0].Set(words4(rec[:4]))
		out[1].SetUint64(rec[4])
	case kindInverse:
		out[0].SetUint64(rec[0])
	default:
		out[0].SetUint64(rec[0])
		out[1].SetUint64(rec[1])
	}
}
