// functionhandler.h -- sampling of the right-hand side, the boundary data and the exact solution
// on a grid; interface of functionhandler.h:27-75.
#ifndef FUNCTIONHANDLER_H
#define FUNCTIONHANDLER_H

#include "multigridsolver2D.h"
#include "multigridsolver3D.h"

typedef real (*Function2D)(real, real);
typedef real (*Function3D)(real, real, real);

class FunctionHandler2D
{
public:
    FunctionHandler2D(Function2D f, Function2D borders, Function2D solution = 0)
        : m_pFunc(f), m_pBord(borders), m_pSol(solution)
    {
    }
    Buffer2D discretize_func(int dimx, int dimy, real dh, const BorderHandler2D &bord);
    Buffer2D discretize_sol(int dimx, int dimy, real dh, const BorderHandler2D &bord);
    bool hasSol() const { return m_pSol != 0; }

private:
    Function2D m_pFunc, m_pBord, m_pSol;
};

class FunctionHandler3D
{
public:
    FunctionHandler3D(Function3D f, Function3D borders, Function3D solution = 0)
        : m_pFunc(f), m_pBord(borders), m_pSol(solution)
    {
    }
    Buffer3D discretize_func(int dimx, int dimy, int dimz, real dh, const BorderHandler3D &bord);
    Buffer3D discretize_sol(int dimx, int dimy, int dimz, real dh, const BorderHandler3D &bord);
    bool hasSol() const { return m_pSol != 0; }

private:
    Function3D m_pFunc, m_pBord, m_pSol;
};

#endif // FUNCTIONHANDLER_H
