// LinearEquationSolverHip::BuildBdcIndex -- the one member of the adapters that reads the mesh (Mesh::_dofOffset), kept in its own
// translation unit: Mesh.hpp of the FEMuS tree pulls in the whole finite-element layer, so a FEMuS build compiles this file with
// the rest of its library, while the interface check of tests/test_adapters_vs_reference_headers.py (no boost in this image)
// covers HipBackend.cpp.  BuildBdcIndex serves LinearEquationSolverPetsc::BuildBdcIndex (LinearEquationSolverPetsc.cpp:53-90).
#include "HipBackend.hpp"
#include "Mesh.hpp"
#include <algorithm>

namespace femus {

void LinearEquationSolverHip::BuildBdcIndex(const std::vector<unsigned>& variable_to_be_solved) {
  // Rows the level solver keeps as identity rows (SetPenalty / ZerosBoundaryResiduals): every owned dof of a variable that is NOT solved for,
  // and of the solved variables the dofs whose boundary flag is below 1.5 (2 = free, 1 = AMR-constrained, 0 = Dirichlet); ascending.
  _bdcIndexIsInitialized = true;
  const int rank = processor_id();
  const Mesh& mesh = *GetMeshFromLinEq();
  std::vector<char> solved(_SolPdeIndex.size(), 0);
  for (unsigned v : variable_to_be_solved) solved[v] = 1;
  _bdcIndex.clear();
  _bdcIndex.reserve(KKoffset[KKIndex.size() - 1][rank] - KKoffset[0][rank]);
  std::vector<int> nodes;
  std::vector<double> flag;
  for (size_t v = 0; v < _SolPdeIndex.size(); v++) {
    const unsigned sol = _SolPdeIndex[v], family = _SolType[sol];
    const unsigned node0 = mesh._dofOffset[family][rank], node1 = mesh._dofOffset[family][rank + 1];
    const int row0 = (int)KKoffset[v][rank];
    if (!solved[v]) {
      for (unsigned k = 0; k < node1 - node0; k++) _bdcIndex.push_back(row0 + (int)k);
      continue;
    }
    // the flag vector of this variable in one transfer instead of one host-synchronous operator()(i) per dof
    nodes.resize(node1 - node0);
    for (unsigned k = 0; k < node1 - node0; k++) nodes[k] = (int)(node0 + k);
    (*_Bdc)[sol]->get(nodes, flag);
    for (unsigned k = 0; k < node1 - node0; k++)
      if (flag[k] < 1.5) _bdcIndex.push_back(row0 + (int)k);
  }
  std::sort(_bdcIndex.begin(), _bdcIndex.end());
}

// Element blocks of the FEMuS_ASM solver from `SetElementBlockNumber` / `SetNumberOfSchurVariables` -- what
// LinearEquationSolverPetscAsm::BuildASMIndex (petsc_asm/LinearEquationSolverPetscAsm.cpp:91-276) and MeshASMPartitioning::DoPartition
// (02_partitioning/MeshASMPartitioning.cpp:89-150) hand PCASM as overlapping subdomains, written from the contract:
//   * the elements this process owns are split by material class (solid = 4, porous = 3, every other flag) and, inside a class, into runs of
//     `_elementBlockNumber` consecutive elements: one block per run, classes in that order;
//   * "non-Schur" variables are all variables of the system except the last `_NSchurVar` entries of the solve list; a block holds their
//     system dofs on every OWNED element around one of its elements (the vertex neighbourhood, or only the face neighbourhood when the
//     first Schur variable is a discontinuous family), and the Schur variables' dofs on its own elements;
//   * every dof once, ascending; dofs of other processes that these elements carry belong to the set as well (PCASM's overlapping set).
// PC_ASM_BASIC corrects the whole overlapping set, so that set is the block of the smoother.
void LinearEquationSolverHipAsm::BuildASMIndex(const std::vector<unsigned>& variable_to_be_solved) {
  const Mesh& mesh = *GetMeshFromLinEq();
  const unsigned rank = processor_id();
  const unsigned first_elem = mesh.GetElementOffset(rank), end_elem = mesh.GetElementOffset(rank + 1);
  const size_t nvar = _SolPdeIndex.size(), nsolve = variable_to_be_solved.size();

  std::vector<char> is_schur(nvar, 0);
  for (size_t k = nsolve - _NSchurVar; k < nsolve; k++) is_schur[variable_to_be_solved[k]] = 1;
  // neighbourhood layer of Elem::GetElementNearElement: 1 = vertex neighbours, 0 = face neighbours (discontinuous Schur variable, family >= 3)
  bool vertex_layer = false;
  if (_NSchurVar != 0) vertex_layer = _SolType[_SolPdeIndex[variable_to_be_solved[nsolve - _NSchurVar]]] < 3;

  // runs of consecutive owned elements per material class
  std::vector<unsigned> of_class[3];
  for (unsigned e = first_elem; e < end_elem; e++) {
    const unsigned flag = mesh.GetElementMaterial(e);
    of_class[flag == 4 ? 0 : flag == 3 ? 1 : 2].push_back(e);
  }
  const size_t run = std::max<size_t>(_elementBlockNumber, 1);

  auto dofs_of = [&](unsigned elem, char schur, std::vector<int>& out) {       // system dofs of the (non-)Schur variables on one element
    for (size_t v = 0; v < nvar; v++) {
      if (is_schur[v] != schur) continue;
      const unsigned sol = _SolPdeIndex[v];
      const unsigned n = mesh.GetElementDofNumber(elem, _SolType[sol]);
      for (unsigned i = 0; i < n; i++) out.push_back((int)GetSystemDof(sol, (unsigned)v, i, elem));
    }
  };

  _blockPtr.assign(1, 0);
  _blockDofs.clear();
  _blockExactCount = 0;
  std::vector<int> last_block_of(end_elem - first_elem, -1);     // block that took an owned element into its neighbourhood last
  std::vector<int> dofs;
  int block = 0;
  for (int cls = 0; cls < 3; cls++) {
    const std::vector<unsigned>& elems = of_class[cls];
    // the blocks of the solid and the porous elements come first and get the exact sub-solve (`_blockTypeRange[1]`, LinearEquationSolverPetscAsm.cpp:298-307)
    if (cls < 2) _blockExactCount += (int)((elems.size() + run - 1) / run);
    for (size_t begin = 0; begin < elems.size(); begin += run, block++) {
      const size_t end = std::min(elems.size(), begin + run);
      dofs.clear();
      for (size_t k = begin; k < end; k++) {
        const unsigned own = elems[k];
        const unsigned nnear = mesh.GetMeshElements()->GetElementNearElementSize(own, vertex_layer);
        for (unsigned j = 0; j < nnear; j++) {
          const unsigned around = mesh.GetMeshElements()->GetElementNearElement(own, j);
          if (around < first_elem || around >= end_elem || last_block_of[around - first_elem] == block) continue;
          last_block_of[around - first_elem] = block;
          dofs_of(around, 0, dofs);
        }
        dofs_of(own, 1, dofs);
      }
      std::sort(dofs.begin(), dofs.end());
      dofs.erase(std::unique(dofs.begin(), dofs.end()), dofs.end());
      _blockDofs.insert(_blockDofs.end(), dofs.begin(), dofs.end());
      _blockPtr.push_back((int)_blockDofs.size());
    }
  }
}

}  // namespace femus
