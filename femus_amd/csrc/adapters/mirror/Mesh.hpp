// the reduced Mesh of this mirror lives in LinearEquationSolver.hpp; this header exists so that sources can include "Mesh.hpp" by
// the name the FEMuS tree uses
#pragma once
#include "LinearEquationSolver.hpp"
