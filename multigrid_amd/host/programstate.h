// programstate.h -- the interactive driver: option parsing, command loop, `solve`.
// Public interface of programstate.h:29-36; commands and options of programstate.cpp:36-53,
// 178-269.  Built twice: with -DBIDIM for the 2D executable, without for the 3D one.
#ifndef PROGRAMSTATE_H
#define PROGRAMSTATE_H

#include "functionhandler.h"
#include "multigridsolver2D.h"
#include "multigridsolver3D.h"
#include "rectangularborderhandler.h"
#include <iosfwd>
#include <string>

class ProgramState
{
public:
    enum SolverMode { Fmg = 0, Smooth = 1, Multigrid = 2 };

    ProgramState(int argc, char **argv);
    void listenCommand();

private:
#ifdef BIDIM
    typedef Buffer2D Buffer;
    typedef RectangularBorderHandler RecBorderHandler;
    typedef MultigridSolver2D Solver;
    typedef FunctionHandler2D FuncHandler;
#else
    typedef Buffer3D Buffer;
    typedef ParallelepipedalBorderHandler RecBorderHandler;
    typedef MultigridSolver3D Solver;
    typedef FunctionHandler3D FuncHandler;
#endif

    typedef void (ProgramState::*Command)(std::istream &);
    struct CommandTableEntry {
        const char *CmdName;
        Command Func;
        const char *Description;
    };
    static const CommandTableEntry CommandTable[];
    static const int CommandCount;

    void setdim(std::istream &params);
    void setsmoothsteps(std::istream &params);
    void setmode(std::istream &params);
    void setvcycles(std::istream &params);
    void setomega(std::istream &params);
    void setiterations(std::istream &params);
    void setfullz(std::istream &params);
    void state(std::istream &params);
    void print(std::istream &params);
    void quit(std::istream &params);
    void help(std::istream &params);
    void solve(std::istream &params);
    void save(std::istream &params);
    void reduce(std::istream &params);
    void prolongate(std::istream &params);
    void helpString();

    // the field a command's `what` argument names, or null after printing `unknown`
    Buffer *pick(const std::string &what, bool announceMissing, const char *unknown);
    Buffer makeBuffer(int w, int h, int d);
    Buffer makeZeroed(int w, int h, int d);

    SolverMode m_curMode;
    int m_dimx, m_dimy, m_dimz;
    int stepA1, stepA2, VCycles, iterations;
    real m_omega;
    bool m_bDisplaySolution, m_bDisplayResidual, m_bDisplayError;
    bool m_bProfilingMode;
    bool m_bFullZ;

    Buffer m_residual, m_error, m_solution, m_targetFunction;
    RecBorderHandler m_handler;
    Solver m_solver;
    FuncHandler m_funcHandler;
};

#endif // PROGRAMSTATE_H
