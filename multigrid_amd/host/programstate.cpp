// programstate.cpp -- option parsing, the command loop and `solve` (programstate.cpp:150-662).
// Prompts, messages and the result table are the reference's, character for character, so that
// scripted sessions replay unchanged; `--fullz` / `setfullz` are additions (3D: relax every z
// plane instead of the first ceil(nz/2), multigridsolver3D.cpp:121).
#include "programstate.h"
#include "auxiliary.h"
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

#ifndef M_PI
#define M_PI 3.141592653589793238462643
#endif

using namespace std;

// ---- test problems (programstate.cpp:55-122) -----------------------------------------------------
// `real` arguments, bodies in double because of M_PI, rounded to `real` on return.
namespace {
#ifdef BIDIM
real sinfunc2(real x, real y)
{
    return -26 * 26 * M_PI * M_PI * sin(26 * M_PI * x) - 50 * 50 * M_PI * M_PI * cos(50 * M_PI * y) - M_PI * M_PI * sin(M_PI * x);
}
real sinfunc2sol(real x, real y) { return sin(M_PI * 26 * x) + cos(M_PI * 50 * y) + sin(M_PI * x); }
#else
real tridimF1(real x, real y, real z) { return -3.0 * M_PI * M_PI * sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z); }
real tridimS1(real x, real y, real z) { return sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z); }
#endif
} // namespace

// field text layout (programstate.cpp:124-148): one line per x, y running along the line
std::ostream &operator<<(std::ostream &os, const BidimArray<real> &m)
{
    for (int i = 0; i < m.width(); ++i) {
        for (int j = 0; j < m.height(); ++j)
            os << m(i, j) << " ";
        os << std::endl;
    }
    return os;
}

std::ostream &operator<<(std::ostream &os, const TridimArray<real> &m)
{
    for (int k = 0; k < m.depth(); ++k) {
        for (int i = 0; i < m.width(); ++i) {
            for (int j = 0; j < m.height(); ++j)
                os << m(i, j, k) << " ";
            os << std::endl;
        }
        os << std::endl;
    }
    return os;
}

const ProgramState::CommandTableEntry ProgramState::CommandTable[] = {
    {"solve", &ProgramState::solve, "solve: Solves the problem with the current parameters"},
    {"quit", &ProgramState::quit, "quit: Exits the program"},
    {"print", &ProgramState::print,
     "print what: Prints some function. \"print err\" prints the error function, \"print sol\" prints the solution, "
     "\"print res\" prints the residuals"},
    {"setdim", &ProgramState::setdim, "setdim dimx dimy: Sets working dimensions"},
    {"setmode", &ProgramState::setmode,
     "setmode mode: Sets the solving mode. \"fmg\" for FMG method, \"mg\" for default multi-grid method, \"sor\" for "
     "simple Successive-Overrelaxation"},
    {"setsmoothsteps", &ProgramState::setsmoothsteps,
     "setsmoothsteps a1 a2: sets the number of pre and post relaxation iteration for FMG and MG method"},
    {"setomega", &ProgramState::setomega, "setomega o: sets the omega parameter for every smoother"},
    {"state", &ProgramState::state, "state: prints the current state of the solver"},
    {"setvcycles", &ProgramState::setvcycles, "setvcycles: set the number of cycles for multigrid and FMG methods"},
    {"setiterations", &ProgramState::setiterations, "setiterations: sets the number of iterations in a mg or fmg solver"},
    {"save", &ProgramState::save, "save what filename: save (sol err res) to the output file \"filename\" "},
    {"reduce", &ProgramState::reduce, "reduce what: reduce (sol err res)"},
    {"prolongate", &ProgramState::prolongate, "prolongate what: prolongate (sol err res)"},
    {"help", &ProgramState::help, "help: lists all the commands"},
    {"setfullz", &ProgramState::setfullz, "setfullz 0|1: (3D) smooth every z plane instead of the first half"},
};
const int ProgramState::CommandCount = sizeof(ProgramState::CommandTable) / sizeof(ProgramState::CommandTable[0]);

namespace {
// argv[i+1] as a C string, or abort like an out-of-range argv access of the reference would
const char *nextArg(int argc, char **argv, int &i)
{
    if (i + 1 >= argc) {
        cout << "Missing value for " << argv[i] << endl;
        exit(1);
    }
    return argv[++i];
}
void require(bool ok, const char *complaint)
{
    if (!ok) {
        cout << complaint << endl;
        exit(1);
    }
}
} // namespace

ProgramState::ProgramState(int argc, char **argv)
    : m_curMode(Fmg), m_dimx(17), m_dimy(17), m_dimz(17), stepA1(3), stepA2(3), VCycles(2), iterations(3), m_omega(1.0),
      m_bDisplaySolution(false), m_bDisplayResidual(false), m_bDisplayError(false), m_bProfilingMode(false),
      m_bFullZ(false),
#ifdef BIDIM
      m_solver("mg_0.cl", m_handler), m_funcHandler(sinfunc2, sinfunc2sol, sinfunc2sol)
#else
      m_solver("mg_1.cl", m_handler), m_funcHandler(tridimF1, tridimS1, tridimS1)
#endif
{
    for (int i = 0; i < argc; ++i) {
        const string opt = argv[i];
        if (opt == "--solver") {
            const string v = nextArg(argc, argv, i);
            if (v == "fmg")
                m_curMode = Fmg;
            else if (v == "jac")
                m_curMode = Smooth;
            else if (v == "mg")
                m_curMode = Multigrid;
            else {
                cout << v << "is not a valid solver type" << endl;
                exit(1);
            }
        } else if (opt == "--dim") {
            m_dimx = atoi(nextArg(argc, argv, i));
            m_dimy = atoi(nextArg(argc, argv, i));
#ifndef BIDIM
            m_dimz = atoi(nextArg(argc, argv, i));
#endif
            require(m_dimx > 0 && m_dimy > 0 && m_dimz > 0, "Dimensions not valid");
        } else if (opt == "--smoothsteps") {
            stepA1 = atoi(nextArg(argc, argv, i));
            stepA2 = atoi(nextArg(argc, argv, i));
            require(stepA1 >= 0 && stepA2 >= 0, "Can not do less than 0 smooth steps");
        } else if (opt == "--mgcycles") {
            VCycles = atoi(nextArg(argc, argv, i));
            require(VCycles >= 0, "Can not do less than 0 V-Cycles");
        } else if (opt == "--displaysol") {
            m_bDisplaySolution = true;
        } else if (opt == "--displayres") {
            m_bDisplayResidual = true;
        } else if (opt == "--displayerr") {
            m_bDisplayError = true;
        } else if (opt == "-h" || opt == "--help") {
            helpString();
        } else if (opt == "--profiling") {
            m_bProfilingMode = true;
        } else if (opt == "--omega") {
            m_omega = atof(nextArg(argc, argv, i));
            require(!(m_omega < 0 || m_omega >= 2), "Invalid omega");
        } else if (opt == "--iterations") {
            iterations = atoi(nextArg(argc, argv, i));
            require(iterations >= 0, "Invalid iterations");
        } else if (opt == "--fullz") {
            m_bFullZ = true;
        } else {
            cout << "Unkown option " << argv[i] << endl;
            helpString();
            abort();
        }
    }
#ifdef BIDIM
    m_dimz = 1;
#endif
}

void ProgramState::listenCommand()
{
    cout << "Real epsilon is: " << std::numeric_limits<real>::epsilon() << endl;
    if (m_bProfilingMode) {
        istringstream none;
        solve(none);
        return;
    }
    for (;;) {
        cout << ">";
        string line, cmd;
        if (!std::getline(cin, line)) // end of input ends the session (the reference spins here)
            return;
        istringstream params(line);
        params >> cmd;
        bool done = false;
        for (int i = 0; i < CommandCount; ++i)
            if (cmd == CommandTable[i].CmdName) {
                (this->*CommandTable[i].Func)(params);
                done = true;
            }
        if (!done)
            cout << "Unknown command: " << cmd << endl;
        if (cmd == "quit")
            return;
    }
}

void ProgramState::setdim(std::istream &params)
{
    int x = 0, y = 0, z = 1;
    params >> x >> y;
#ifndef BIDIM
    params >> z;
#endif
    if (params.fail() || x < 0 || y < 0 || z < 0)
        cout << "Invalid dimensions" << endl;
    else
        m_dimx = x, m_dimy = y, m_dimz = z;
}

void ProgramState::setsmoothsteps(istream &params)
{
    int a1, a2;
    params >> a1 >> a2;
    if (params.fail() || a1 < 0 || a2 < 0)
        cout << "Invalid parameters" << endl;
    else
        stepA1 = a1, stepA2 = a2;
}

void ProgramState::setmode(istream &params)
{
    string mode;
    params >> mode;
    if (params.fail())
        cout << "Invalid mode" << endl; // and once more below, as programstate.cpp:343-352 does
    if (mode == "fmg")
        m_curMode = Fmg;
    else if (mode == "sor")
        m_curMode = Smooth;
    else if (mode == "mg")
        m_curMode = Multigrid;
    else
        cout << "Invalid mode" << endl;
}

void ProgramState::setvcycles(istream &params)
{
    int v;
    params >> v;
    if (params.fail() || v < 0)
        cout << "Invalid number of VCycles" << endl;
    else
        VCycles = v;
}

void ProgramState::setomega(std::istream &params)
{
    ::real omega;
    params >> omega;
    if (params.fail() || omega < 0 || omega > 2.0)
        cout << "Invalid omega" << endl;
    else
        m_omega = omega;
}

void ProgramState::setiterations(std::istream &params)
{
    int iters;
    params >> iters;
    if (params.fail() || iters < 0)
        cout << "Invalid iterations" << endl;
    else
        iterations = iters;
}

void ProgramState::setfullz(std::istream &params)
{
    int on;
    params >> on;
    if (params.fail() || (on != 0 && on != 1))
        cout << "Invalid parameters" << endl;
    else
        m_bFullZ = on != 0;
}

void ProgramState::state(std::istream &)
{
    static const char *const names[] = {"FMG", "Jacobi Smoother", "Multigrid"};
    cout << "Current solver: " << names[m_curMode] << endl
         << "Dimension: " << m_dimx << "x" << m_dimy << endl
         << "Pre smooth steps: " << stepA1 << " Post smooth steps: " << stepA2 << endl
         << "VCycles: " << VCycles << " Omega: " << m_omega << endl;
}

ProgramState::Buffer *ProgramState::pick(const std::string &what, bool announceMissing, const char *unknown)
{
    struct Named {
        const char *key;
        Buffer *buf;
        const char *missing;
    } const table[] = {{"sol", &m_solution, "No solution available"},
                       {"res", &m_residual, "No residuals available"},
                       {"err", &m_error, "No error available"},
                       {"func", &m_targetFunction, "No function available"}};
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i)
        if (what == table[i].key) {
            if (!announceMissing || table[i].buf->isInitialized())
                return table[i].buf;
            cout << table[i].missing << endl;
            return 0;
        }
    cout << unknown << endl;
    return 0;
}

void ProgramState::print(std::istream &params)
{
    string what;
    params >> what;
    if (Buffer *b = pick(what, true, "Print what?"))
        cout << b->read(m_solver.queue()) << endl;
}

void ProgramState::quit(std::istream &) { cout << "Quitting..." << endl; }

void ProgramState::help(std::istream &)
{
    for (int i = 0; i < CommandCount; ++i)
        cout << CommandTable[i].Description << endl;
}

void ProgramState::save(istream &params)
{
    string what, filename;
    params >> what >> filename;
    if (params.fail())
        cout << "Save what where?" << endl;
    Buffer *arg = pick(what, false, "Save what?");
    if (!arg)
        return;
    if (!arg->isInitialized()) {
        cout << "Data not initialized yet" << endl;
        return;
    }
    if (filename.size() > 4 && filename.substr(filename.size() - 3) == "bmp") {
#ifdef BIDIM
        toBitmap(*arg, m_solver.queue(), filename.c_str());
#else
        cout << "Saving to bitmap not supported in 3D." << endl;
#endif
    } else {
        ofstream outfile(filename.c_str());
        if (!outfile)
            cout << "Can not open: " << filename << endl;
        else
            outfile << arg->read(m_solver.queue()) << endl;
    }
}

ProgramState::Buffer ProgramState::makeBuffer(int w, int h, int d)
{
#ifdef BIDIM
    (void)d;
    return Buffer(w, h);
#else
    return Buffer(w, h, d);
#endif
}

ProgramState::Buffer ProgramState::makeZeroed(int w, int h, int d)
{
#ifdef BIDIM
    (void)d;
    return Buffer::empty(w, h, m_solver.queue());
#else
    return Buffer::empty(w, h, d, m_solver.queue());
#endif
}

#ifdef BIDIM
static int depthOf(const Buffer2D &) { return 1; }
#else
static int depthOf(const Buffer3D &b) { return b.depth(); }
#endif

void ProgramState::reduce(istream &params)
{
    string what;
    params >> what;
    if (Buffer *p = pick(what, true, "Reduce what?")) {
        Buffer s = makeBuffer(p->width() / 2 + 1, p->height() / 2 + 1, depthOf(*p) / 2 + 1);
        m_solver.restrict(s, *p);
        *p = s;
    }
}

void ProgramState::prolongate(istream &params)
{
    string what;
    params >> what;
    if (Buffer *p = pick(what, true, "Prolongate what?")) {
        Buffer s = makeBuffer(p->width() * 2 - 1, p->height() * 2 - 1, depthOf(*p) * 2 - 1);
        m_solver.prolongate(s, *p);
        *p = s;
    }
}

void ProgramState::solve(std::istream &)
{
    const real dh = 1.0 / (m_dimx - 1);
#ifdef BIDIM
    m_targetFunction = m_funcHandler.discretize_func(m_dimx, m_dimy, dh, m_handler);
#else
    m_targetFunction = m_funcHandler.discretize_func(m_dimx, m_dimy, m_dimz, dh, m_handler);
    m_solver.setFullZRange(m_bFullZ);
#endif
    m_solution = makeZeroed(m_dimx, m_dimy, m_dimz);
    m_residual = makeZeroed(m_dimx, m_dimy, m_dimz);
    m_error = makeZeroed(m_dimx, m_dimy, m_dimz);

    const clock_t startTimer = clock();

    switch (m_curMode) {
    case Fmg:
        m_solution = m_solver.fmg(m_targetFunction, m_omega, stepA1, stepA2, VCycles, iterations);
        break;
    case Smooth:
        m_solver.smoother_iterate(m_solution, m_targetFunction, m_omega, stepA1);
        break;
    case Multigrid:
        m_solver.mg(m_solution, m_targetFunction, m_omega, stepA1, stepA2, VCycles, iterations);
        break;
    }
    m_solver.zero_out(m_solution);

    if (m_funcHandler.hasSol()) {
#ifdef BIDIM
        Buffer exact = m_funcHandler.discretize_sol(m_dimx, m_dimy, dh, m_handler);
#else
        Buffer exact = m_funcHandler.discretize_sol(m_dimx, m_dimy, m_dimz, dh, m_handler);
#endif
        m_error = Difference(m_solution, exact, m_solver.queue());
    }

    m_solver.compute_residuals(m_residual, m_solution, m_targetFunction);
    m_solver.wait();

    const clock_t endTimer = clock();

    cout << "Time\t\t\tL2Err\t\t\tLInfErr\t\t\tL2Res\t\t\tLinfRes\t\t\t" << endl;
    cout << (double)(endTimer - startTimer) / CLOCKS_PER_SEC << "\t\t\t" << L2Norm(m_error, m_solver.queue()) << "\t\t\t"
         << LInfNorm(m_error, m_solver.queue()) << "\t\t\t" << L2Norm(m_residual, m_solver.queue()) << "\t\t\t"
         << LInfNorm(m_residual, m_solver.queue()) << endl;
    // --displaysol / --displayres / --displayerr are parsed and, as in the reference, never acted on
}

void ProgramState::helpString()
{
    cout << "Usage: multigrid [options] " << endl;
    cout << "Valid options are: " << endl;
    cout << "\t \"--solver solvertype\": set the solver type. 'fmg' for a Fast multigrid solver, 'jac' for the Jacobi "
            "method, 'mg' for the simple multigrid method"
         << endl
         << "\t \"--dim dimx dimy\": set the dimension of the grid to be used" << endl
         << "\t \"--smoothsteps a1 a2\": sets the number of steps of pre and post smoothing. For simple jacobi method only "
            "the first value is used"
         << endl
         << "\t \"--mgcycles v\": sets the number of v-cycles to be used (1= V Cycle, 2 = W Cycle)" << endl
         << "\t \"--omega o\": sets the omega parameter" << endl
         << "\t \"--profiling\": profiling mode: print solution data and exits immediatly" << endl
         << "\t \"--help\" \"-h\": prints this help" << endl
         << endl;
}
