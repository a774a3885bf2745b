// LinearEquationSolverHip::BuildBdcIndex -- the one member of the adapters that reads the mesh (Mesh::_dofOffset), kept in its own
// translation unit: Mesh.hpp of the FEMuS tree pulls in the whole finite-element layer, so a FEMuS build compiles this file with
// the rest of its library, while the interface check of tests/test_adapters_vs_reference_headers.py (no boost in this image)
// covers HipBackend.cpp.  Same statement order as LinearEquationSolverPetsc::BuildBdcIndex (LinearEquationSolverPetsc.cpp:53-90).
#include "HipBackend.hpp"
#include "Mesh.hpp"
#include <algorithm>

namespace femus {

void LinearEquationSolverHip::BuildBdcIndex(const std::vector<unsigned>& variable_to_be_solved) {
  _bdcIndexIsInitialized = true;
  const int p = processor_id();
  _bdcIndex.resize(KKoffset[KKIndex.size() - 1][p] - KKoffset[0][p]);
  std::vector<bool> ThisSolutionIsIncluded(_SolPdeIndex.size(), false);
  for (unsigned iind = 0; iind < variable_to_be_solved.size(); iind++) ThisSolutionIsIncluded[variable_to_be_solved[iind]] = true;
  unsigned count0 = 0;
  for (unsigned k = 0; k < _SolPdeIndex.size(); k++) {
    const unsigned indexSol = _SolPdeIndex[k];
    const unsigned soltype = _SolType[indexSol];
    const unsigned first = GetMeshFromLinEq()->_dofOffset[soltype][p], last = GetMeshFromLinEq()->_dofOffset[soltype][p + 1];
    if (!ThisSolutionIsIncluded[k]) {
      for (unsigned inode = first; inode < last; inode++) _bdcIndex[count0++] = KKoffset[k][p] + (inode - first);
      continue;
    }
    // the flag vector of this variable in one transfer instead of one host-synchronous operator()(i) per dof
    std::vector<int> idx(last - first);
    for (unsigned inode = first; inode < last; inode++) idx[inode - first] = (int)inode;
    std::vector<double> flag;
    (*_Bdc)[indexSol]->get(idx, flag);
    for (unsigned inode = first; inode < last; inode++)
      if (flag[inode - first] < 1.5) _bdcIndex[count0++] = KKoffset[k][p] + (inode - first);     // 2 = free, 1 = AMR-constrained, 0 = Dirichlet
  }
  _bdcIndex.resize(count0);
  std::sort(_bdcIndex.begin(), _bdcIndex.end());
}

// Element blocks of the FEMuS_ASM solver from `SetElementBlockNumber` / `SetNumberOfSchurVariables`: the statement order of
// LinearEquationSolverPetscAsm::BuildASMIndex (petsc_asm/LinearEquationSolverPetscAsm.cpp:91-276) with MeshASMPartitioning::DoPartition
// (02_partitioning/MeshASMPartitioning.cpp:89-150) inlined -- runs of `_elementBlockNumber` consecutive owned elements per material class
// (solid 4, porous 3, everything else), non-Schur variables on the elements around the block (vertex neighbours unless the Schur
// variable is discontinuous: "FastVankaBlock"), Schur variables on the block's own elements.  The reference hands PCASM the overlapping
// index set of a block as its subdomain (PC_ASM_BASIC: the whole subdomain is corrected); that set is the block here.
void LinearEquationSolverHipAsm::BuildASMIndex(const std::vector<unsigned>& variable_to_be_solved) {
  const Mesh* msh = GetMeshFromLinEq();
  bool FastVankaBlock = true;
  if (_NSchurVar != 0)
    FastVankaBlock = (_SolType[_SolPdeIndex[variable_to_be_solved[variable_to_be_solved.size() - _NSchurVar]]] < 3) ? false : true;   // NFE_FAMS_C_ZERO_LAGRANGE
  const unsigned iproc = processor_id();
  const unsigned DofOffset = KKoffset[0][iproc];
  const unsigned DofOffsetSize = KKoffset[KKIndex.size() - 1][iproc] - KKoffset[0][iproc];
  std::vector<unsigned> indexb(DofOffsetSize, DofOffsetSize);
  const unsigned ElemOffset = msh->GetElementOffset(iproc), ElemOffsetp1 = msh->GetElementOffset(iproc + 1);
  const unsigned ElemOffsetSize = ElemOffsetp1 - ElemOffset;
  std::vector<unsigned> indexci(ElemOffsetSize), indexc(ElemOffsetSize, ElemOffsetSize);
  // ---- DoPartition ----
  std::vector<std::vector<unsigned> > block_elements;
  {
    const unsigned block_size[3] = {_elementBlockNumber, _elementBlockNumber, _elementBlockNumber};
    const unsigned flag_block[3] = {4, 3, 2};
    for (unsigned iMaterial = 0; iMaterial < 3; iMaterial++) {
      std::vector<unsigned> mine;
      for (unsigned iel = ElemOffset; iel < ElemOffsetp1; iel++) {
        const unsigned flag_mat = msh->GetElementMaterial(iel);
        const bool here = iMaterial < 2 ? flag_mat == flag_block[iMaterial] : (flag_mat != flag_block[0] && flag_mat != flag_block[1]);
        if (here) mine.push_back(iel);
      }
      for (size_t k = 0; k < mine.size(); k += block_size[iMaterial])
        block_elements.emplace_back(mine.begin() + k, mine.begin() + std::min(mine.size(), k + block_size[iMaterial]));
    }
  }
  std::vector<bool> ThisVaribaleIsNonSchur(_SolPdeIndex.size(), true);
  for (unsigned iind = variable_to_be_solved.size() - _NSchurVar; iind < variable_to_be_solved.size(); iind++)
    ThisVaribaleIsNonSchur[variable_to_be_solved[iind]] = false;
  _blockPtr.assign(1, 0);
  _blockDofs.clear();
  for (size_t vb_index = 0; vb_index < block_elements.size(); vb_index++) {
    std::vector<int> over;
    unsigned Csize = 0;
    for (size_t kel = 0; kel < block_elements[vb_index].size(); kel++) {
      const unsigned iel = block_elements[vb_index][kel];
      for (unsigned j = 0; j < msh->GetMeshElements()->GetElementNearElementSize(iel, !FastVankaBlock); j++) {
        const unsigned jel = msh->GetMeshElements()->GetElementNearElement(iel, j);
        if (jel < ElemOffset || jel >= ElemOffsetp1 || indexc[jel - ElemOffset] != ElemOffsetSize) continue;
        indexci[Csize] = jel - ElemOffset;
        indexc[jel - ElemOffset] = Csize++;
        for (unsigned indexSol = 0; indexSol < _SolPdeIndex.size(); indexSol++) {       // non-Schur variables of the elements around
          if (!ThisVaribaleIsNonSchur[indexSol]) continue;
          const unsigned SolPdeIndex = _SolPdeIndex[indexSol], SolType = _SolType[SolPdeIndex];
          for (unsigned jj = 0; jj < msh->GetElementDofNumber(jel, SolType); jj++) {
            const unsigned kkdof = GetSystemDof(SolPdeIndex, indexSol, jj, jel);
            if (kkdof - DofOffset < DofOffsetSize && indexb[kkdof - DofOffset] != DofOffsetSize) continue;
            if (kkdof - DofOffset < DofOffsetSize) indexb[kkdof - DofOffset] = (unsigned)over.size();
            over.push_back((int)kkdof);
          }
        }
      }
      for (unsigned indexSol = 0; indexSol < _SolPdeIndex.size(); indexSol++) {         // Schur variables of the block's own elements
        if (ThisVaribaleIsNonSchur[indexSol]) continue;
        const unsigned SolPdeIndex = _SolPdeIndex[indexSol], SolType = _SolType[SolPdeIndex];
        for (unsigned ii = 0; ii < msh->GetElementDofNumber(iel, SolType); ii++) {
          const unsigned kkdof = GetSystemDof(SolPdeIndex, indexSol, ii, iel);
          if (kkdof - DofOffset < DofOffsetSize && indexb[kkdof - DofOffset] != DofOffsetSize) continue;
          if (kkdof - DofOffset < DofOffsetSize) indexb[kkdof - DofOffset] = (unsigned)over.size();
          over.push_back((int)kkdof);
        }
      }
    }
    for (int d : over)
      if ((unsigned)d - DofOffset < DofOffsetSize) indexb[(unsigned)d - DofOffset] = DofOffsetSize;
    for (unsigned i = 0; i < Csize; i++) indexc[indexci[i]] = ElemOffsetSize;
    std::sort(over.begin(), over.end());
    over.erase(std::unique(over.begin(), over.end()), over.end());
    _blockDofs.insert(_blockDofs.end(), over.begin(), over.end());
    _blockPtr.push_back((int)_blockDofs.size());
  }
}

}  // namespace femus
