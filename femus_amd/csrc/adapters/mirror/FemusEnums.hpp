// Enumerations of the FEMuS algebra interface as the real tree declares them (global scope, same enumerators in the same order, so
// the values agree): src/00_enums/algebra/SolverPackageEnum.hpp, 01_matrices/ParalleltypeEnum.hpp,
// 02_preconditioners/PrecondtypeEnum.hpp, 03_solvers_with_preconditioner/{SolvertypeEnum,MgTypeEnum,LinearEquationSolverEnum}.hpp.
// HIP_SOLVERS is the one enumerator a FEMuS maintainer adds; here it takes the place of TRILINOS_SOLVERS' successor.
#pragma once
enum SolverPackage { PETSC_SOLVERS = 0, TRILINOS_SOLVERS, HIP_SOLVERS, INVALID_SOLVER_PACKAGE };
enum ParallelType { AUTOMATIC = 0, SERIAL, PARALLEL, GHOSTED, INVALID_PARALLELIZATION };
enum PreconditionerType {
  IDENTITY_PRECOND = 0, JACOBI_PRECOND, BLOCK_JACOBI_PRECOND, SOR_PRECOND, SSOR_PRECOND, EISENSTAT_PRECOND, ASM_PRECOND, ASM_ADDITIVE_PRECOND,
  ASM_MULTIPLICATIVE_PRECOND, CHOLESKY_PRECOND, ICC_PRECOND, ILU_PRECOND, LU_PRECOND, USER_PRECOND, SHELL_PRECOND, AMG_PRECOND,
  INVALID_PRECONDITIONER, MG_PRECOND, SLU_PRECOND, MLU_PRECOND, ULU_PRECOND, MCC_PRECOND, FIELDSPLIT_PRECOND, FIELDSPLIT_ADDITIVE_PRECOND,
  FIELDSPLIT_MULTIPLICATIVE_PRECOND, FIELDSPLIT_SYMMETRIC_MULTIPLICATIVE_PRECOND, FIELDSPLIT_SCHUR_PRECOND, LSC_PRECOND
};
enum SolverType {
  CG = 0, CGN, CGS, CR, QMR, TCQMR, TFQMR, BICG, BICGSTAB, MINRES, FGMRES, GMRES, LGMRES, LSQR, JACOBI, SOR_FORWARD, SOR_BACKWARD, SSOR,
  RICHARDSON, CHEBYSHEV, LUMP, INVALID_SOLVER, PREONLY
};
enum MgType { F_CYCLE = 0, V_CYCLE, M_CYCLE };
enum MgSmootherType { FULL = 0, MULTIPLICATIVE, ADDITIVE, KASKADE };
enum LinearEquationSolverType { FEMuS_DEFAULT = 0, FEMuS_ASM, FEMuS_FIELDSPLIT };
#ifndef LSOLVER
#define LSOLVER HIP_SOLVERS          // FemusConfig.hpp defines it under HAVE_PETSC (FemusConfig.hpp.in:62-63); here the backend at hand
#endif
