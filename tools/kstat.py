import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Name"]:
        print(f'{row["Name"].split("(")[0][-42:]:42s} calls={row["Calls"]} avg_us={float(row["AverageNs"])/1e3:.1f}')
