"""Result tables of the evaluation drivers -- the four printers tools/train_net.py and the evaluation helper of the
reference call (cubercnn/vis/logperf.py:9-117; `import cubercnn.vis.logperf as utils_logperf`, tools/train_net.py:56).
Same function names, arguments and table layouts (tabulate); the ANSI colouring of termcolor is applied only when that
package is installed."""
import logging

from tabulate import tabulate

logger = logging.getLogger(__name__)

try:
    from termcolor import colored
except ImportError:                       # not in this image: plain text
    def colored(text, *_a, **_k):
        return text


def _grid(rows, headers):
    return tabulate(rows, headers=headers, tablefmt="grid", numalign="left", stralign="center")


def print_ap_category_histogram(dataset, results):
    """per-category AP2D / AP3D of one dataset, three (category, AP2D, AP3D) triples per row (logperf.py:9-43)"""
    n_cols = 9
    flat = [v for cat, out in results.items() for v in (cat, out["AP2D"], out["AP3D"])]
    flat.extend([None] * (n_cols - (len(flat) % n_cols)))
    rows = [flat[i:i + n_cols] for i in range(0, len(flat), n_cols)]
    table = tabulate(rows, headers=["category", "AP2D", "AP3D"] * (n_cols // 3), tablefmt="pipe", numalign="left",
                     stralign="center")
    logger.info("Performance for each of {} categories on {}:\n".format(len(results), dataset) + colored(table, "cyan"))
    return table


def print_ap_analysis_histogram(results):
    """per-dataset AP at the IoU thresholds and depth ranges (logperf.py:46-66)"""
    keys = ["AP2D", "AP3D", "AP3D@15", "AP3D@25", "AP3D@50", "AP3D-N", "AP3D-M", "AP3D-F"]
    rows = [[name, m["iters"]] + [m[k] for k in keys] for name, m in results.items()]
    table = _grid(rows, ["Dataset", "#iters"] + keys)
    logger.info("Per-dataset performance analysis on test set:\n" + colored(table, "cyan"))
    return table


def print_ap_dataset_histogram(results):
    """per-dataset AP2D / AP3D (logperf.py:69-90)"""
    rows = [[name, m["iters"], m["AP2D"], m["AP3D"]] for name, m in results.items()]
    table = _grid(rows, ["Dataset", "#iters", "AP2D", "AP3D"])
    logger.info("Per-dataset performance on test set:\n" + colored(table, "cyan"))
    return table


def print_ap_omni_histogram(results):
    """the Omni3D / Omni3D_In / Omni3D_Out summary that is compared across methods (logperf.py:93-117)"""
    rows = [[name, m["iters"], m["AP2D"], m["AP3D"]] for name, m in results.items()]
    table = _grid(rows, ["Dataset", "#iters", "AP2D", "AP3D"])
    logger.info("Omni3D performance on test set. The numbers below should be used to compare to others approaches on "
                "Omni3D, such as Cube R-CNN")
    logger.info("Performance on Omni3D:\n" + colored(table, "magenta"))
    return table
