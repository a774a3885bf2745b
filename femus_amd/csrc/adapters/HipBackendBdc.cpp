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

}  // namespace femus
