// main.cpp -- entry point of multigrid2D / multigrid3D (main.cpp:14-47): build the program state
// from the command line (without argv[0]) and run the command loop; report exceptions and exit 1.
#include "programstate.h"
#include <iostream>

int main(int argc, char **argv)
{
    try {
        ProgramState ps(argc - 1, argv + 1);
        ps.listenCommand();
    } catch (const mgrt::Error &e) {
        std::cout << "Runtime error in " << e.what() << " code: " << e.err() << std::endl << "Aborting.." << std::endl;
        return 1;
    } catch (const std::exception &e) {
        std::cout << "Exception: " << e.what() << std::endl << "Aborting..." << std::endl;
        return 1;
    } catch (...) {
        std::cout << "Unknown error" << std::endl << std::endl << "Aborting..." << std::endl;
        return 1;
    }
    return 0;
}
