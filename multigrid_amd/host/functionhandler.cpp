// functionhandler.cpp -- grid sampling (functionhandler.cpp:25-101).
// Coordinates are (real)i/(n-1); inner cells get f*dh*dh, Dirichlet cells the boundary value,
// Neumann cells boundary*dh, outer cells are left as allocated (0 here).
#include "functionhandler.h"

Buffer2D FunctionHandler2D::discretize_func(int dimx, int dimy, real dh, const BorderHandler2D &bord)
{
    BidimArray<real> buf(dimx, dimy);
    for (int i = 0; i < dimx; ++i)
        for (int j = 0; j < dimy; ++j) {
            const real x = (real)(i) / (dimx - 1), y = (real)(j) / (dimy - 1);
            const BorderHandler2D::CellType t = bord.cellType(i, j, dimx, dimy);
            if (t == BorderHandler2D::CellInner)
                buf(i, j) = m_pFunc(x, y) * dh * dh;
            else if (t == BorderHandler2D::CellDirichlet)
                buf(i, j) = m_pBord(x, y);
            else if (t == BorderHandler2D::CellNeumann)
                buf(i, j) = m_pBord(x, y) * dh;
        }
    return Buffer2D(dimx, dimy, buf.data(), Buffer2D::ReadOnly);
}

Buffer2D FunctionHandler2D::discretize_sol(int dimx, int dimy, real, const BorderHandler2D &bord)
{
    BidimArray<real> buf(dimx, dimy);
    for (int i = 0; i < dimx; ++i)
        for (int j = 0; j < dimy; ++j)
            buf(i, j) = bord.cellType(i, j, dimx, dimy) == BorderHandler2D::CellOuter
                            ? (real)0
                            : m_pSol((real)(i) / (dimx - 1), (real)(j) / (dimy - 1));
    return Buffer2D(dimx, dimy, buf.data(), Buffer2D::ReadOnly);
}

Buffer3D FunctionHandler3D::discretize_func(int dimx, int dimy, int dimz, real dh, const BorderHandler3D &bord)
{
    TridimArray<real> buf(dimx, dimy, dimz);
    for (int k = 0; k < dimz; ++k)
        for (int j = 0; j < dimy; ++j)
            for (int i = 0; i < dimx; ++i) {
                const real x = (real)(i) / (dimx - 1), y = (real)(j) / (dimy - 1), z = (real)(k) / (dimz - 1);
                const BorderHandler3D::CellType t = bord.cellType(i, j, k, dimx, dimy, dimz);
                if (t == BorderHandler3D::CellInner)
                    buf(i, j, k) = m_pFunc(x, y, z) * dh * dh;
                else if (t == BorderHandler3D::CellDirichlet)
                    buf(i, j, k) = m_pBord(x, y, z);
                else if (t == BorderHandler3D::CellNeumann)
                    buf(i, j, k) = m_pBord(x, y, z) * dh;
            }
    return Buffer3D(dimx, dimy, dimz, buf.data(), Buffer3D::ReadOnly);
}

Buffer3D FunctionHandler3D::discretize_sol(int dimx, int dimy, int dimz, real, const BorderHandler3D &bord)
{
    TridimArray<real> buf(dimx, dimy, dimz);
    for (int k = 0; k < dimz; ++k)
        for (int j = 0; j < dimy; ++j)
            for (int i = 0; i < dimx; ++i)
                buf(i, j, k) = bord.cellType(i, j, k, dimx, dimy, dimz) == BorderHandler3D::CellOuter
                                   ? (real)0
                                   : m_pSol((real)(i) / (dimx - 1), (real)(j) / (dimy - 1), (real)(k) / (dimz - 1));
    return Buffer3D(dimx, dimy, dimz, buf.data(), Buffer3D::ReadOnly);
}
